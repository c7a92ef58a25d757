#!/usr/bin/env python3
"""Small-batch latency of the drop-in env path (the reference itself runs ONE environment): wall time of one 180 s
env step (1 800 RK4 sub-steps, full scenario) for N = 1 ... 65 536 spacecraft through the Python binding, and of
leoPowerAttEnv.reset() with the pooled device handle.  Usage (GPU box): python tools/latency.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM, GRAV_PM_J2  # noqa: E402
from basilisk_env_amd.envs import leoPowerAttEnv  # noqa: E402
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config  # noqa: E402
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch  # noqa: E402

# LATENCY_GRAV=j2: J2 gravity (the kernels the A/B variant libraries carry); LATENCY_ONLY_STEP=1: the batch table only
GRAV = GRAV_PM_J2 if os.environ.get("LATENCY_GRAV") == "j2" else GRAV_PM
out = {"env_step_ms": {}, "gravity": "j2" if GRAV == GRAV_PM_J2 else "pm", "lib": os.environ.get("BSKGPU_LIB", "libbskgpu.so")}
for n in (1, 2, 63, 64, 1024, 4096, 8192, 65536):
    cfg = default_config(3, GRAV)
    cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 3, seed=1))
    act = np.zeros(n, np.int32)
    p.step(act, 1800)
    p.get_obs()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        p.step(act, 1800)
        p.get_obs()
        ts.append(time.perf_counter() - t0)
    out["env_step_ms"][n] = round(min(ts) * 1e3, 3)
    p.close()
if os.environ.get("LATENCY_ONLY_STEP") == "1":
    print(json.dumps(out))
    sys.exit(0)
env = leoPowerAttEnv()
env.seed(1)
t0 = time.perf_counter()
env.reset()
first = time.perf_counter() - t0
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    env.reset()
    ts.append(time.perf_counter() - t0)
t0 = time.perf_counter()
for _ in range(5):
    env.step(0)
out["single_env"] = {"first_reset_ms": round(first * 1e3, 3), "pooled_reset_ms": round(min(ts) * 1e3, 3),
                     "step_ms": round((time.perf_counter() - t0) / 5 * 1e3, 3)}
env.close()
# the batched sibling as an RL loop drives it: LeoPowerAttVecEnv.step() = H2D of the actions, the step kernel, D2H of
# obs / reward / done, the (N,5,1) observation array and the per-env infos (wall time per 180 s env step)
from basilisk_env_amd.envs import LeoPowerAttVecEnv  # noqa: E402
out["vec_env_step_ms"] = {}
for n, pool in ((1024, 0), (65536, 0), (65536, 4096)):
    venv = LeoPowerAttVecEnv(n, device_reset_pool=pool)
    venv.reset()
    acts = np.zeros(n, dtype=np.int64)
    venv.step(acts)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        venv.step(acts)
        ts.append(time.perf_counter() - t0)
    out["vec_env_step_ms"]["%d%s" % (n, "_device_reset" if pool else "")] = round(min(ts) * 1e3, 3)
    venv.close()
print(json.dumps(out))
