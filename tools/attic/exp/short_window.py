"""The driver's bench form times 20 launches between two synchronisations: where do the microseconds beyond 20 kernels go?
usage: tools/exp/short_window.py [spin]   (spin: hipSetDeviceFlags(hipDeviceScheduleSpin) before anything touches the GPU)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] in ("spin", "yield", "block"):
    hip = ctypes.CDLL("libamdhip64.so")
    flag = {"spin": 1, "yield": 2, "block": 4}[sys.argv[1]]
    print("hipSetDeviceFlags(%d) ->" % flag, hip.hipSetDeviceFlags(flag))
import torch
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = 65536
cfg = default_config(4, GRAV_PM_J2)
p = BatchedPropagator(cfg, n); p.reset(sample_ic_batch(n, 4, seed=0))
act = torch.zeros(n, dtype=torch.int32, device="cuda"); ptr = act.data_ptr()
def run(steps, sync):
    ts = []
    for _ in range(300):
        for _ in range(5): p.step_device(ptr, 1)
        p.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): p.step_device(ptr, 1)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        ts.append((t2 - t0, t1 - t0))
    a = np.array(ts) * 1e6
    return np.median(a[:, 0]), np.median(a[:, 1])
for steps in (1, 20, 200):
    for name, s in (("torch.cuda.synchronize", torch.cuda.synchronize), ("bsk_sync (hipStreamSynchronize)", p.sync)):
        tot, enq = run(steps, s)
        print("%4d steps, %-32s: %.1f us total (%.2f per step), host enqueue %.1f us" % (steps, name, tot, tot / steps, enq))
