"""Spherical-harmonic gravity field data for the batched propagator (BASELINE config 5).

The reference's only harmonics hook is ``useSphericalHarmonicsGravityModel(file, degree)``
(opNav_models/BSK_OpNavDynamics.py:211-214), which reads a GGM coefficient file from Basilisk's
supportData.  That file is not available here, so the benchmark field is synthetic
(SURVEY.md §8(d)): the real C20 plus Kaula-rule noise.
"""
import numpy as np

from .config import CBAR_20


def sh_index(l, m):
    """Packed index of (l, m), 0 <= m <= l, used by ``bsk_set_gravity_sh`` and the oracle."""
    return l * (l + 1) // 2 + m


def sh_size(degree):
    return (degree + 1) * (degree + 2) // 2


def synthetic_sh_coefficients(degree=70, seed=70):
    """Normalised C/S up to ``degree``: C00 = 1, degree-1 terms zero, real C20, everything else
    Kaula-rule noise N(0, (1e-5/l^2)^2) from PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = np.zeros(sh_size(degree))
    s = np.zeros(sh_size(degree))
    c[sh_index(0, 0)] = 1.0
    for l in range(2, degree + 1):
        sig = 1e-5 / (l * l)
        for m in range(l + 1):
            c[sh_index(l, m)] = rng.normal(0.0, sig)
            s[sh_index(l, m)] = rng.normal(0.0, sig) if m > 0 else 0.0
    c[sh_index(2, 0)] = CBAR_20
    return c, s


def zonal_j2_only(degree=2):
    """Field with only C20 — must reproduce the closed-form J2 model."""
    c = np.zeros(sh_size(degree))
    s = np.zeros(sh_size(degree))
    c[sh_index(0, 0)] = 1.0
    c[sh_index(2, 0)] = CBAR_20
    return c, s
