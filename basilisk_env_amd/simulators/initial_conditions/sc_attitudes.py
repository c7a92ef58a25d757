"""Attitude initial conditions (mirrors reference ``simulators/initial_conditions/sc_attitudes.py``)."""
import numpy as np


def random_tumble(maxSpinRate=0.001):
    """sigma_BN ~ U(0,1)^3, omega_BN_B ~ U(-max, max)^3 from the legacy numpy global RNG
    (reference sc_attitudes.py:3-13; pinned by tests/golden/ic_random_tumble.json)."""
    sigma_bn = np.random.uniform(0, 1.0, [3, ])
    omega_bn = np.random.uniform(-maxSpinRate, maxSpinRate, [3, ])
    return sigma_bn, omega_bn


def static_inertial():
    """Zero attitude and rate (reference sc_attitudes.py:15-23)."""
    return np.zeros([3, ]), np.zeros([3, ])


def random_tumble_batch(n, rng, maxSpinRate=0.001):
    """Batched form on a ``numpy.random.Generator``; arrays of shape (n, 3)."""
    return rng.uniform(0, 1.0, (n, 3)), rng.uniform(-maxSpinRate, maxSpinRate, (n, 3))
