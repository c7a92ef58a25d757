#!/usr/bin/env python3
"""Role probe of the three-wave form: probe libraries variants/probe_tri_role{1,2,3}.so (make -C basilisk_env_amd/csrc probe-libs
PROBES="tri_role1:-DBSK_PROBE_TRI_ROLE=1 tri_role2:-DBSK_PROBE_TRI_ROLE=2 tri_role3:-DBSK_PROBE_TRI_ROLE=3"), one ROLE each -
rotational wave (1), FSW + environment wave (2), translational wave (3): per tick, the cycles of its tick loop, the cycles of them
spent at workgroup barriers, and the cycles re-reading the exchange (dynamics halves) or running the FSW chain (environment wave);
mean over the batch's workgroups.  Each library is loaded in a child process.  usage: tri_roles.py N [K [SEED]]      (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
role = int(os.environ.get("TRI_ROLE_CHILD", "0"))
names = ("rotational wave   ", "FSW + environment ", "translational wave")
third = ("re-reading the exchange", "in the FSW chain", "re-reading the exchange")
if role == 0:
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for r in (1, 2, 3):
        env = dict(os.environ, TRI_ROLE_CHILD=str(r), BSKGPU_LIB=os.path.join(here, "basilisk_env_amd", "variants", "probe_tri_role%d.so" % r))
        res = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
        sys.stdout.write(res.stdout if res.returncode == 0 else "role %d failed: %s\n" % (r, res.stderr[-500:]))
    sys.exit(0)
os.environ["BSKGPU_TRI"] = "1"
cfg = default_config(4, GRAV_PM_J2); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=seed))
act = np.zeros(n, np.int32)
walls = []
for _ in range(6):
    t0 = time.perf_counter(); p.step(act, K); p.sync(); walls.append(time.perf_counter() - t0)
wall = min(walls[1:])
w = [int(v) for v in p.debug_words()]
f = lambda s: float(np.mean([((v >> s) & 0xFFFFF) * 64 for v in w]))     # noqa: E731
loop, bar, x = f(0), f(20), f(40)
k = role - 1
if role == 1:
    print("%s  n %d K %d seed %d: wall of one launch %.3f ms = %.0f ns per tick (%.2f counter ticks per ns)" % (p.kernel_info()["name"], n, K, seed, wall * 1e3, wall / K * 1e9, loop / (wall * 1e9)))
print("  %s loop %7.0f cycles/tick   at barriers %6.0f (%4.1f %%)   %s %6.0f (%4.1f %%)   otherwise (issuing its own instructions) %6.0f" %
      (names[k], loop / K, bar / K, 100 * bar / max(loop, 1), third[k], x / K, 100 * x / max(loop, 1), (loop - bar - x) / K))
p.close()
