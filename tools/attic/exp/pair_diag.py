"""Where do the pair form and the single-wave form part?  Reproduces tests/test_gpu_pair.py's full-nosun case and prints the
first call / rows / envs that differ.  usage: tools/exp/pair_diag.py [tri]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
form = "BSKGPU_TRI" if len(sys.argv) > 1 else "BSKGPU_PAIR"
def make(cfg, n, on):
    os.environ["BSKGPU_PAIR"] = "0"; os.environ["BSKGPU_TRI"] = "0"
    os.environ[form] = "1" if on else "0"
    p = BatchedPropagator(cfg, n)
    return p
n, n_rw = 333, 4
cfg = default_config(n_rw, GRAV_PM_J2)
cfg.flags |= FLAG_POWER | FLAG_DRAG | FLAG_DESAT
cfg.fsw_lag, cfg.nav_lag = 1, 1
cfg.base_density, cfg.scale_height = 1e-9, 100e3
ic = sample_ic_batch(n, n_rw, seed=17)
ic[12:12 + n_rw, ::5] *= 4.0
for ks in ((1, 16, 20, 37), (1,) * 74):
    a, b = make(cfg, n, False), make(cfg, n, True)
    print(a.kernel_info()["name"][:60], "|", b.kernel_info()["name"][:60])
    a.reset(ic); b.reset(ic)
    rng = np.random.default_rng(4)
    tick = 0
    for call, k in enumerate(ks):
        act = rng.integers(0, 3, n).astype(np.int32) if len(ks) == 4 else np.full(n, 0, np.int32)
        a.step(act, k); b.step(act, k); tick += k
        sa, sb = a.get_state(), b.get_state()
        bad = np.argwhere(sa != sb)
        if len(bad):
            rows = sorted(set(bad[:, 0].tolist())); envs = sorted(set(bad[:, 1].tolist()))
            print("call %d k %d (tick %d): %d entries differ; rows %s; envs %s ... ; actions of those %s" % (call, k, tick, len(bad), rows, envs[:12], act[envs[:12]].tolist()))
            r0, e0 = bad[0]
            print("   first: row %d env %d  %r vs %r" % (r0, e0, sa[r0, e0], sb[r0, e0]))
            break
    else:
        print("no difference over", ks[:4], "...")
    a.close(); b.close()
