#!/usr/bin/env python3
"""Where LeoPowerAttVecEnv.step() of 65 536 spacecraft spends its wall time beyond the kernel."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd.envs import LeoPowerAttVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = LeoPowerAttVecEnv(n, n_rw=4, seed=1, device_reset_pool=4096)
env.reset()
act = np.zeros(n, np.int64)
for _ in range(3): env.step(act)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
print("step wall ms: min %.3f median %.3f" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
p = env.propagator
ts = []
for _ in range(5):
    t0 = time.perf_counter(); p.step(act.astype(np.int32), 1800); p.sync(); ts.append(time.perf_counter() - t0)
print("propagator.step+sync ms: min %.3f" % (min(ts) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): env.step(act)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
