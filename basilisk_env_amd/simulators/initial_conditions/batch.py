"""Batched synthetic random-orbit initial conditions on a numpy Generator.

Same distributions as the reference's per-episode sampler (``set_ICs``,
simulators/leoPowerAttitudeSimulator.py:119-193): orbit ``leo_orbit.sampled_400km`` (:25-40),
tumble ``sc_attitudes.random_tumble(1e-5)`` (:124), disturbance torque ``2e-4 * N(0,1)^3`` (:151-152,
295 — the un-normalised vector, as the reference uses it), wheel speeds U(-800, 800) RPM (:155),
battery U(8, 20) W h (:167).
"""
import numpy as np

from ..dynamics.config import RPM
from ..dynamics.propagator import pack_ic
from . import leo_orbit, sc_attitudes


def sample_ic_batch(n, n_rw, rng=None, seed=0, disturbance_magnitude=2e-4):
    """-> SoA block ``[n_fields, n]`` ready for ``BatchedPropagator.reset``."""
    if rng is None:
        rng = np.random.Generator(np.random.PCG64(seed))
    _, rN, vN = leo_orbit.sample_batch(n, rng)
    sigma, omega = sc_attitudes.random_tumble_batch(n, rng, maxSpinRate=0.00001)
    lext = disturbance_magnitude * rng.standard_normal((n, 3))
    wheels = rng.uniform(-800, 800, (n, n_rw)) * RPM if n_rw else None
    charge = rng.uniform(8. * 3600., 20. * 3600., n)
    return pack_ic(n_rw, rN, vN, sigma, omega, wheelSpeeds=wheels, lext=lext, charge=charge)
