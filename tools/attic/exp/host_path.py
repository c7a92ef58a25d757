"""The boundary with HOST buffers on both sides at K = 1 (bsk_step with host actions + bsk_get_obs into page-locked host arrays, one
synchronisation per step): microseconds per step, for a library given by BSKGPU_LIB.  usage: host_path.py [N_ENVS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = BatchedPropagator(default_config(4, GRAV_PM_J2), n)
p.reset(sample_ic_batch(n, 4, seed=0))
a = np.zeros(n, np.int32)
best = 1e9
for rep in range(5):
    for _ in range(20):
        p.step(a, 1); p.get_obs(copy=False)
    t0 = time.perf_counter()
    for _ in range(300):
        p.step(a, 1); p.get_obs(copy=False)
    best = min(best, (time.perf_counter() - t0) / 300)
print("n %d: %.1f us per step with host buffers (%.3g env-steps/s)" % (n, best * 1e6, n / best))
p.close()
