#!/usr/bin/env python3
"""A few launches of the full-scenario kernel for counter passes: python3 tools/exp/step_once.py N K REPS (form by BSKGPU_PAIR / BSKGPU_TRI)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n, K, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
act = np.zeros(n, np.int32)
for _ in range(reps):
    t0 = time.perf_counter(); p.step(act, K); p.sync(); dt = time.perf_counter() - t0
print(p.kernel_info()["name"], "n", n, "K", K, "last launch %.3f ms" % (dt * 1e3))
p.close()
