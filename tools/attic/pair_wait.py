import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
for _ in range(3):
    p.step(np.zeros(n, np.int32), K); p.sync()
m = p.debug_words()
wb = (m & np.uint64(0x1FFFFF)).astype(np.float64) * 16; wa = ((m >> np.uint64(21)) & np.uint64(0x1FFFFF)).astype(np.float64) * 16; ch = ((m >> np.uint64(42)) & np.uint64(0x1FFFFF)).astype(np.float64) * 16
print("envs", n, "per launch of", K, "ticks, mean / max over waves [kcycles]: D waits at B %.0f / %.0f, at A %.0f / %.0f; F chain total %.0f / %.0f" % (wb.mean() / 1e3, wb.max() / 1e3, wa.mean() / 1e3, wa.max() / 1e3, ch.mean() / 1e3, ch.max() / 1e3))
