import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
# exchange probe of the three-wave form (library built with -DBSK_TRI_DEBUG): per launch of K ticks and wave, how many consumes
# found their value not yet published, how many re-reads that took, and the cycles spent re-reading
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
os.environ["BSKGPU_TRI"] = "1"
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
import time
for _ in range(3):
    t0 = time.perf_counter()
    p.step(np.zeros(n, np.int32), K); p.sync()
    wall = time.perf_counter() - t0
print("wall of the last launch %.3f ms (%.0f ns per tick)" % (wall * 1e3, wall / K * 1e9))
arr = p.debug_words()
miss, spin, cyc = (arr & np.uint64(0xFFFF)).astype(float), ((arr >> np.uint64(16)) & np.uint64(0xFFFF)).astype(float), (arr >> np.uint64(32)).astype(float) * 64
print("%-14s envs %d K %d: consumes early %.0f (of %d), re-reads %.0f, kcycles re-reading %.0f (mean over waves; max %.0f)" % (sys.argv[3] if len(sys.argv) > 3 else "", n, K, miss.mean(), 4 * K, spin.mean(), cyc.mean() / 1e3, cyc.max() / 1e3))
