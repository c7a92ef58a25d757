import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
cfg = default_config(3, GRAV_PM); cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, 1); p.reset(sample_ic_batch(1, 3, seed=1)); act = np.zeros(1, np.int32)
for _ in range(3): p.step(act, 1800); p.get_obs(); p.get_state()
def t(f, n=20):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return min(ts)*1e3, sorted(ts)[len(ts)//2]*1e3
print("step+sync  %.3f %.3f" % t(lambda: (p.step(act,1800), p.sync())))
print("step K=1   %.3f %.3f" % t(lambda: (p.step(act,1), p.sync())))
print("get_obs    %.3f %.3f" % t(p.get_obs))
print("get_state  %.3f %.3f" % t(p.get_state))
print("sync       %.3f %.3f" % t(p.sync))
