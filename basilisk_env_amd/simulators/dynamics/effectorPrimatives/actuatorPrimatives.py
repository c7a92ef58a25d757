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
