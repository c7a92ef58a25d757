#!/usr/bin/env python3
"""Where the drop-in env's step() spends its wall time: the launch, the read-back, the Python around them."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd.envs import leoPowerAttEnv
env = leoPowerAttEnv(); env.seed(1); env.reset()
for _ in range(3): env.step(0)
ts = []
for a in (0, 1, 0, 2, 0, 0, 1, 0, 0, 0):
    t0 = time.perf_counter(); env.step(a); ts.append(time.perf_counter() - t0)
print("env.step wall ms: min %.3f median %.3f" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
sim = env.simulator
p = sim.propagator if hasattr(sim, "propagator") else None
if p is not None:
    act = np.zeros(1, np.int32)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); p.step(act, 1800); p.sync(); ts.append(time.perf_counter() - t0)
    print("propagator step+sync ms: min %.3f" % (min(ts) * 1e3), p.kernel_info()["name"])
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); p.get_obs_state(); ts.append(time.perf_counter() - t0)
    print("get_obs_state ms: min %.3f" % (min(ts) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): env.step(0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
