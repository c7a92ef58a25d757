"""Orbit initial conditions (mirrors reference ``simulators/initial_conditions/leo_orbit.py``).

``elem2rv`` restates the closed-form classical-elements -> (r, v) conversion of Basilisk's
``orbitalMotion.elem2rv`` (Schaub & Junkins), the only Basilisk function the reference calls
here (leo_orbit.py:21,38).
"""
import math

import numpy as np

MU_EARTH = 0.3986004415E+15
D2R = math.pi / 180.0


class ClassicElements(object):
    a = None
    e = None
    i = None
    Omega = None
    omega = None
    f = None


def elem2rv(mu, oe):
    """Classical elements -> inertial position/velocity.  Accepts scalars or equal-shape arrays
    in ``oe`` and returns arrays of shape (..., 3) (or (3,) for scalars)."""
    a, e, i, Om, om, f = (np.asarray(x, dtype=float) for x in (oe.a, oe.e, oe.i, oe.Omega, oe.omega, oe.f))
    p = a * (1.0 - e * e)
    r = p / (1.0 + e * np.cos(f))
    theta = om + f
    ct, st, cO, sO, ci, si = np.cos(theta), np.sin(theta), np.cos(Om), np.sin(Om), np.cos(i), np.sin(i)
    rN = np.stack([r * (cO * ct - sO * st * ci), r * (sO * ct + cO * st * ci), r * (st * si)], axis=-1)
    h = np.sqrt(mu * p)
    A, B = st + e * np.sin(om), ct + e * np.cos(om)
    vN = np.stack([-mu / h * (cO * A + sO * B * ci), -mu / h * (sO * A - cO * B * ci), -mu / h * (-B * si)], axis=-1)
    return rN, vN


def rv2elem(mu, rN, vN):
    """Inverse of :func:`elem2rv` for non-circular, inclined orbits (round-trip tests)."""
    rN, vN = np.asarray(rN, float), np.asarray(vN, float)
    r, v2 = np.linalg.norm(rN), float(np.dot(vN, vN))
    h = np.cross(rN, vN)
    n = np.cross([0.0, 0.0, 1.0], h)
    evec = ((v2 - mu / r) * rN - np.dot(rN, vN) * vN) / mu
    oe = ClassicElements()
    oe.e = np.linalg.norm(evec)
    oe.a = 1.0 / (2.0 / r - v2 / mu)
    oe.i = math.acos(h[2] / np.linalg.norm(h))
    oe.Omega = math.atan2(n[1], n[0]) % (2 * math.pi)
    oe.omega = math.acos(np.clip(np.dot(n, evec) / (np.linalg.norm(n) * oe.e), -1, 1))
    if evec[2] < 0:
        oe.omega = 2 * math.pi - oe.omega
    oe.f = math.acos(np.clip(np.dot(evec, rN) / (oe.e * r), -1, 1))
    if np.dot(rN, vN) < 0:
        oe.f = 2 * math.pi - oe.f
    return oe


def inclined_circular_300km():
    """Inclined circular LEO (reference leo_orbit.py:6-23)."""
    oe = ClassicElements()
    oe.a = 6371 * 1000.0 + 300. * 1000
    oe.e = 0.0
    oe.i = 45.0 * D2R
    oe.Omega = 0.0
    oe.omega = 0.0
    oe.f = 0.0
    rN, vN = elem2rv(MU_EARTH, oe)
    return oe, rN, vN


def sampled_400km():
    """Randomly sampled LEO (reference leo_orbit.py:25-40): a = 6371 km + 500 km, e~U(0,.05),
    i~U(-90,90) deg, Omega, omega, f ~U(0,360) deg, drawn from the legacy numpy global RNG in
    the reference's order (size-1 arrays, like the reference)."""
    oe = ClassicElements()
    oe.a = 6371 * 1000.0 + 500. * 1000
    oe.e = np.random.uniform(0, 0.05, 1)
    oe.i = np.random.uniform(-90 * D2R, 90 * D2R, 1)
    oe.Omega = np.random.uniform(0 * D2R, 360 * D2R, 1)
    oe.omega = np.random.uniform(0 * D2R, 360 * D2R, 1)
    oe.f = np.random.uniform(0 * D2R, 360 * D2R, 1)
    rN, vN = elem2rv(MU_EARTH, oe)
    return oe, rN.reshape(3), vN.reshape(3)


def sample_batch(n, rng):
    """Batched form of :func:`sampled_400km` on a ``numpy.random.Generator``."""
    oe = ClassicElements()
    oe.a = np.full(n, 6371 * 1000.0 + 500. * 1000)
    oe.e = rng.uniform(0, 0.05, n)
    oe.i = rng.uniform(-90 * D2R, 90 * D2R, n)
    oe.Omega = rng.uniform(0, 360 * D2R, n)
    oe.omega = rng.uniform(0, 360 * D2R, n)
    oe.f = rng.uniform(0, 360 * D2R, n)
    rN, vN = elem2rv(MU_EARTH, oe)
    return oe, rN, vN
