#!/usr/bin/env python3
"""Experiment: one batch of 65 536 spacecraft stepped (K = 1) as 1 / 2 / 4 / 8 concurrent launches on as many streams of ONE
card (ShardedPropagator(devices=[0]*s)).  The K = 1 launch is a latency chain (launch, loads, 0.9 us of arithmetic, stores,
release); independent chains on separate hardware queues can overlap.  Wall time per env step of the whole batch."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.sharded import ShardedPropagator
from basilisk_env_amd.simulators.dynamics import default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
cfg = default_config(4, GRAV_PM_J2)
ic = sample_ic_batch(n, 4, seed=0)
out = {}
for s in (1, 2, 4, 8):
    sp = ShardedPropagator(cfg, n, devices=[0] * s)
    sp.reset(ic)
    acts = [torch.zeros(hi - lo, dtype=torch.int32, device="cuda") for lo, hi in sp.ranges]
    ptrs = [a.data_ptr() for a in acts]
    torch.cuda.synchronize()
    for _ in range(300):
        sp.step_device(ptrs, K)
    for p in sp.shards:
        p.sync()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            sp.step_device(ptrs, K)
        for p in sp.shards:
            p.sync()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None else min(best, dt)
    out[s] = round(best * 1e6, 3)
    print("streams %d: %.3f us per step of %d envs  (%.3g env-steps/s)" % (s, best * 1e6, n, n / best), flush=True)
    sp.close()
print(json.dumps(out))
