"""A few launches of the rollout kernel (bsk_step_n, T = 541, K = 1, 65 536 spacecraft, constant action) for a counter pass:
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 tools/exp/rollout_once.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd import _hip
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n, T = 65536, 541
p = BatchedPropagator(default_config(4, GRAV_PM_J2), n)
p.reset(sample_ic_batch(n, 4, seed=0))
ob = _hip.DeviceBuffer(8 * 5 * n * T, 0); rw = _hip.DeviceBuffer(8 * n * T, 0); wy = _hip.DeviceBuffer(n * T, 0)
for _ in range(3):
    p.step_n(T, 1, None, 0, ob.ptr, rw.ptr, wy.ptr)
p.sync()
print(p.kernel_info())
p.close()
