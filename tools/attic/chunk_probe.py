import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
# chunk probe of the single-wave full-scenario kernel (library built with -DBSK_PROBES=1 -DBSK_PROBE_CHUNK=1|2|3, BSKGPU_LIB):
# the share of the tick loop's cycles one part of every chunk of <= 10 ticks takes
#   usage: BSKGPU_LIB=.../probe_chunkN.so tools/chunk_probe.py [envs] [K] [label] [action]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
os.environ["BSKGPU_TRI"] = "0"; os.environ["BSKGPU_PAIR"] = "0"
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
a = np.full(n, act, np.int32) if act >= 0 else np.random.default_rng(0).integers(0, 3, n).astype(np.int32)
for _ in range(3):
    p.step(a, K); p.sync()
arr = p.debug_words()
part, loop = (arr & np.uint64(0xFFFFFFFF)).astype(float) * 16, (arr >> np.uint64(32)).astype(float) * 16
print("%-10s envs %d K %d action %d: part %.0f cycles per chunk of 10 ticks, loop %.0f cycles per tick, share %.4f (mean over %d waves; min %.4f max %.4f)"
      % (sys.argv[3] if len(sys.argv) > 3 else "", n, K, act, part.mean() / (K / 10), loop.mean() / K, (part / loop).mean(), len(arr), (part / loop).min(), (part / loop).max()))
