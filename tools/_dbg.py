import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
import numpy as np, torch
def P(*a):
    print(*a, file=sys.stderr, flush=True)
from basilisk_env_amd.envs import LeoPowerAttVecEnv
from basilisk_env_amd._lib import GRAV_PM_J2
n=64
env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM_J2, step_duration=2.0, seed=7, device_reset_pool=32)
P("env ok")
ob = env.reset(); P("reset ok")
v = env.propagator.device_views(); P("views", list(v))
for k in ("obs","reward","reason","done_mask","state","terminal_obs","episodes"):
    t = torch.as_tensor(v[k], device="cuda"); P(k, t.shape, t.dtype)
tv = env._torch_views(); P("tviews ok")
a = torch.zeros(n, dtype=torch.int32, device="cuda")
r = env.step_tensors(a); P("step_tensors ok")
torch.cuda.synchronize(); P("sync ok", r[0].shape)
x = r[0].cpu(); P("cpu ok")
d = torch.from_dlpack(v["obs"]); P("dlpack ok", d.shape)
del d; import gc; gc.collect(); P("dlpack del ok")
env.close(); P("close ok")
del env, tv, r, v, t, x; gc.collect(); P("del ok")
