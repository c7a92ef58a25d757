import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
for _ in range(3):
    p.step(np.zeros(n, np.int32), 600); p.sync()
m = p.debug_words()
dur, hid = (m & np.uint64(0xFFFFFFFF)).astype(np.int64), (m >> np.uint64(32)).astype(np.int64)
print("envs", n, "D-wave durations [kcycles] percentiles 0/25/50/75/90/100:", np.percentile(dur, [0, 25, 50, 75, 90, 100]).astype(int))
simd, cu, xcc, se = (hid >> 4) & 3, (hid >> 8) & 15, (hid >> 16) & 15, (hid >> 13) & 7
slow = dur > 1.3 * np.median(dur)
print("  slow waves:", int(slow.sum()), "by simd", collections.Counter(simd[slow].tolist()), "all by simd", collections.Counter(simd.tolist()))
print("  slow by xcc", collections.Counter(xcc[slow].tolist()))
percu = collections.Counter(zip(xcc.tolist(), se.tolist(), cu.tolist()))
print("  D waves per CU histogram", collections.Counter(percu.values()))
slowcu = collections.Counter(zip(xcc[slow].tolist(), se[slow].tolist(), cu[slow].tolist()))
print("  slow D waves per CU histogram", collections.Counter(slowcu.values()), "CUs with slow waves", len(slowcu))
