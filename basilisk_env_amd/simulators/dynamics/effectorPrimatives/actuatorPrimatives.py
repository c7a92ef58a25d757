"""Actuator presets of the scenario, as plain data for the batched propagator.

Mirrors reference ``simulators/dynamics/effectorPrimatives/actuatorPrimatives.py``:
``balancedHR16Triad`` (:7-63, three orthogonal Honeywell HR16 wheels, balanced model,
maxMomentum 50 N m s) and adds the 4-wheel pyramid of
``simulators/opNav_models/BSK_OpNavDynamics.py:269-293`` (elevation 40 deg, azimuth
45/135/225/315 deg), the only 4-wheel definition in the reference.
"""
import math
from collections import namedtuple

import numpy as np

RPM = 2.0 * math.pi / 60.0

# Basilisk simIncludeRW 'Honeywell_HR16' preset at maxMomentum = 50 N m s: Omega_max 6000 RPM,
# u_max 0.2 N m, u_min 1e-5 N m, Coulomb friction 5e-4 N m (cross-checked against the FSW-side
# constants the reference states: wheelJs = 50/(6000 RPM), uMax 0.2 —
# opNav_models/BSK_OpNavFsw.py:425,432).
HONEYWELL_HR16 = {
    "Omega_max": 6000.0 * RPM,
    "maxMomentum": 50.0,
    "u_max": 0.200,
    "u_min": 0.00001,
    "fCoulomb": 0.0005,
}

Wheel = namedtuple("Wheel", "gsHat_B Js Omega u_max u_min fCoulomb")


def _wheel(gs, omega_rpm):
    gs = np.asarray(gs, dtype=float)
    p = HONEYWELL_HR16
    return Wheel(gs, p["maxMomentum"] / p["Omega_max"], omega_rpm * RPM, p["u_max"], p["u_min"], p["fCoulomb"])


def balancedHR16Triad(useRandom=False, randomBounds=(-400, 400)):
    """Three orthogonal HR16 wheels (reference actuatorPrimatives.py:7-63).  With ``useRandom``
    the wheel speeds are drawn from the legacy numpy global RNG exactly like the reference."""
    if useRandom:
        speeds = np.random.uniform(randomBounds[0], randomBounds[1], 3)
    else:
        speeds = np.array([500.0, 500.0, 500.0])
    return [_wheel(ax, s) for ax, s in zip(np.eye(3), speeds)]


def balancedHR16Pyramid(wheelSpeedsRPM=(0.0, 0.0, 0.0, 0.0)):
    """Four-wheel pyramid, gsHat = M3(-az) M2(el) [1,0,0] = [cos az cos el, sin az cos el, sin el]
    with el = 40 deg, az = 45/135/225/315 deg (reference BSK_OpNavDynamics.py:278-291).

    The four axes are built from ONE quadrant's components with explicit signs, so the set is
    exactly symmetric in floating point and sum(g g^T) is exactly diagonal (this is what lets the
    propagator pick its diagonal-inertia kernel); each component equals the per-angle cos/sin
    value to 1 ulp."""
    el, az = 40.0 * math.pi / 180.0, 45.0 * math.pi / 180.0
    cx, cy, cz = math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)
    n = math.sqrt(cx * cx + cy * cy + cz * cz)
    cx, cy, cz = cx / n, cy / n, cz / n
    signs = ((1, 1), (-1, 1), (-1, -1), (1, -1))
    return [_wheel([sx * cx, sy * cy, cz], s) for (sx, sy), s in zip(signs, wheelSpeedsRPM)]


# Basilisk simIncludeThruster 'MOOG_Monarc_1' preset: MaxThrust 0.9 N, MinOnTime 0.02 s.
MOOG_MONARC_1 = {"MaxThrust": 0.9, "MinOnTime": 0.020}

Thruster = namedtuple("Thruster", "pos_B dir_B MaxThrust MinOnTime")


def idealMonarc1Octet():
    """Eight ADCS thrusters with MOOG Monarc-1 attributes at the reference's locations/directions
    (reference actuatorPrimatives.py:66-161: two clusters at y = -/+1.206 m, z = +/-0.85245 m, firing
    along the four (+/-1, +/-1, 0)/sqrt(2) diagonals)."""
    x, y, z = 3.874945160902288e-2, 1.206182747348013, 0.85245
    x2 = 3.8749451609022656e-2
    location = [[x, -y, z], [x, -y, -z], [-x2, -y, z], [-x2, -y, -z],
                [-x, y, z], [-x, y, -z], [x2, y, z], [x2, y, -z]]
    a, b = 0.7071067811865476, 0.7071067811865475
    direction = [[-a, b, 0.0], [-a, b, 0.0], [b, a, 0.0], [b, a, 0.0],
                 [a, -b, 0.0], [a, -b, 0.0], [-b, -a, 0.0], [-b, -a, 0.0]]
    p = MOOG_MONARC_1
    return [Thruster(np.array(r), np.array(g), p["MaxThrust"], p["MinOnTime"]) for r, g in zip(location, direction)]
