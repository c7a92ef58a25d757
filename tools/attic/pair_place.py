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
p.step(np.zeros(n, np.int32), 200); p.sync()
m = p.debug_words()
d, f = (m & np.uint64(0xFFFFFFFF)).astype(np.int64), (m >> np.uint64(32)).astype(np.int64)
def key(h): return ((h >> 16) & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3)   # xcc, se, sh, cu, simd
occ = collections.defaultdict(list)
for i in range(len(m)):
    occ[key(int(d[i]))].append("D"); occ[key(int(f[i]))].append("F")
hist = collections.Counter("".join(sorted(v)) for v in occ.values())
print("envs", n, "SIMDs used", len(occ), dict(hist))
cus = collections.Counter(k[:4] for k in occ)
print("CUs used", len(cus), "waves per CU histogram", collections.Counter(sum(len(occ[k]) for k in occ if k[:4] == c) for c in cus))
