"""A step + a batch-scalars request per iteration, for `rocprofv3 --kernel-trace` (tools/round.sh: kt_stats_<n>): the durations of
stats_kernel and stats_join_kernel (bsk_aux.hip) at one batch size; with `fused` the step launches form the wave sums themselves
(bsk_set_step_stats) and a request is the join kernel alone.
usage: stats_trace.py N_ENVS [ITERATIONS [fused]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd import _hip
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
p = BatchedPropagator(default_config(4, GRAV_PM_J2), n)
p.reset(sample_ic_batch(n, 4, seed=1))
if len(sys.argv) > 3 and sys.argv[3] == "fused":
    p.set_step_stats(True)
act = _hip.DeviceBuffer(4 * n, 0)
_hip.check(_hip.runtime().hipMemsetAsync(act.ptr, 0, 4 * n, None), "hipMemsetAsync")
_hip.stream_sync(0)
for _ in range(iters):
    p.step_device(act.ptr, 1)
    p.batch_stats_device()
p.sync()
s, d = p.batch_stats()
rew = p.get_obs()[1]
print("n", n, "iterations", iters, "sum", s, "host sum", float(rew.sum()), "done", d)
p.close()
