#!/usr/bin/env python3
"""Pair form of the step kernel against the single-wave form on the same inputs (GPU box).  BSKGPU_LIB may point at a variant
library.  Prints max deviations; exits non-zero on a mismatch beyond the last bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2  # noqa: E402
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config  # noqa: E402
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch  # noqa: E402


def make(cfg, n, pair):
    os.environ["BSKGPU_PAIR"] = "1" if pair else "0"
    p = BatchedPropagator(cfg, n)
    del os.environ["BSKGPU_PAIR"]
    return p


bad = 0
for level, flags in (("power", FLAG_POWER), ("full-nosun", FLAG_POWER | FLAG_DRAG | FLAG_DESAT), ("full", FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT)):
    for n_rw in (4, 3):
        for n in (64, 100, 1000):
            cfg = default_config(n_rw, GRAV_PM_J2)
            cfg.flags |= flags
            ic = sample_ic_batch(n, n_rw, seed=5)
            a, b = make(cfg, n, False), make(cfg, n, True)
            a.reset(ic); b.reset(ic)
            rng = np.random.default_rng(1)
            worst = 0.0
            for k in (1, 16, 20, 37, 3, 180, 1):
                act = rng.integers(0, 3, n).astype(np.int32)
                a.step(act, k); b.step(act, k)
                sa, sb = a.get_state(), b.get_state()
                oa, ob = a.get_obs(), b.get_obs()
                scale = np.maximum(np.abs(sa).max(axis=1, keepdims=True), 1e-300)
                dev = float((np.abs(sa - sb) / scale).max())
                worst = max(worst, dev)
                same_int = all(np.array_equal(x, y) for x, y in zip(a.get_counters(), b.get_counters())) and np.array_equal(oa[2], ob[2]) and np.array_equal(oa[3], ob[3])
                if not same_int or dev > (0.0 if level != "full" else 1e-13) or np.abs(oa[0] - ob[0]).max() > (0.0 if level != "full" else 1e-12):
                    bad += 1
                    print("MISMATCH", level, n_rw, n, k, dev, np.abs(oa[0] - ob[0]).max(), same_int)
            print("%-10s n_rw %d n %4d  max rel state dev %.2e  kernel %s" % (level, n_rw, n, worst, b.kernel_info()["name"]), flush=True)
            a.close(); b.close()
sys.exit(1 if bad else 0)
