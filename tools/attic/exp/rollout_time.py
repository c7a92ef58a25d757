"""Open-loop rollouts (bsk_step_n) against one launch per env step: wall time per env step at K sub-steps, T steps per launch.
usage: rollout_time.py [N_ENVS] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from basilisk_env_amd import _hip
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = default_config(4, GRAV_PM_J2)
p = BatchedPropagator(cfg, n)
p.reset(sample_ic_batch(n, 4, seed=0))
Tmax = 541
act = _hip.DeviceBuffer(4 * n * Tmax, 0); _hip.check(_hip.runtime().hipMemsetAsync(act.ptr, 0, 4 * n * Tmax, None), "memset")
ob = _hip.DeviceBuffer(8 * 5 * n * Tmax, 0); rw = _hip.DeviceBuffer(8 * n * Tmax, 0); wy = _hip.DeviceBuffer(n * Tmax, 0)
_hip.stream_sync(0)
def clock(fn, reps):
    fn(); p.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    p.sync(); return (time.perf_counter() - t0) / reps
single = clock(lambda: p.step_device(act.ptr, K), 5000 if K == 1 else 50)
print("n %d K %d: one launch per env step %.2f us -> %.3g env-steps/s" % (n, K, single * 1e6, n / single))
for T in (2, 10, 100, 541):
    for name, a in (("constant action", None), ("actions [T][N]", act.ptr)):
        dt = clock(lambda: p.step_n(T, K, a, 0, ob.ptr, rw.ptr, wy.ptr), max(3, 2000 // T))
        print("  T %3d %-16s %.2f us per env step (%.1f us per launch) -> %.3g env-steps/s, x%.2f; history %.0f GB/s" % (T, name, dt / T * 1e6, dt * 1e6, n * T / dt, single / (dt / T), (49 + (4 if a else 0)) * n * T / dt / 1e9))
dt = clock(lambda: p.step_n(100, K, None, 0, None, None, None), 20)
print("  T 100 no history       %.2f us per env step" % (dt / 100 * 1e6))
print(p.kernel_info())
p.close()
