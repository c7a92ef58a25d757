"""The device-resident loop for the copy tracer (tools/round.sh: rocprofv3 --memory-copy-trace): an on-GPU policy drives
LeoPowerAttVecEnv through reset_tensors / step_tensors for 200 steps.  Everything before the marker line (construction: the IC
pool is drawn on the GPU, the constants are uploaded once) may copy; between the two library counter readings nothing may."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basilisk_env_amd.envs import LeoPowerAttVecEnv
from basilisk_env_amd.simulators.dynamics import BatchedPropagator
n = 65536
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env = LeoPowerAttVecEnv(n, n_rw=4, step_duration=0.1, seed=0, device_reset_pool=4096, device_sampler=True, stream=s.cuda_stream)
    w = torch.randn(5, 3, dtype=torch.float64, device="cuda")
    env._torch_views()
    torch.cuda.synchronize()
    c0 = BatchedPropagator.debug_counters()
    ob = env.reset_tensors()
    for _ in range(200):
        ob, rew, done, info = env.step_tensors((ob.reshape(n, 5) @ w).argmax(dim=1))
    c1 = BatchedPropagator.debug_counters()
    torch.cuda.synchronize()
print("library copies / syncs inside the loop:", c1[0] - c0[0], c1[1] - c0[1], "finished episodes:", int(info["episodes"].sum()))
env.close()
