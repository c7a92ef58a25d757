"""Batch statistics (bsk_get_batch_stats) of a fixed sequence of steps, as hex floats: run once per library (BSKGPU_LIB) and diff."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
for n in (1, 63, 64, 65, 1000, 65536, 70001):
    cfg = default_config(4, GRAV_PM_J2)
    cfg.max_length = 3
    p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=3))
    rng = np.random.default_rng(1)
    for k in (1, 5, 2, 1):
        p.step(rng.integers(0, 3, n).astype(np.int32), k)
        s, d = p.batch_stats()
        obs, rew, done, why = p.get_obs()
        print(n, k, float(s).hex(), d, int((why != 0).sum()), "%.3e" % abs(s - float(np.sum(rew))))
    p.close()
