#!/usr/bin/env python3
"""Experiment: the on-GPU RL step (policy -> step_tensors -> reward accumulation) captured ONCE into a HIP graph through
torch.cuda.CUDAGraph and replayed - the step kernel's arguments do not change from launch to launch (counters, clocks and
episode state live in device memory), so a replay is a valid step.  K = 1, 65 536 spacecraft."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basilisk_env_amd.envs import LeoPowerAttVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = 2000 if K == 1 else 100
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env = LeoPowerAttVecEnv(n, n_rw=4, step_duration=0.1 * K, seed=0, device_reset_pool=4096, device_sampler=True, stream=s.cuda_stream)
    ob = env.reset_tensors()
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(5, 3, dtype=torch.float64, device="cuda", generator=g)
    ret = torch.zeros(n, dtype=torch.float64, device="cuda")
    obs_view = env._torch_views()["obs_n51"]

    def one():
        act = (obs_view.reshape(n, 5) @ w).argmax(dim=1).to(torch.int32)
        ob2, rew, done, _ = env.step_tensors(act)
        ret.add_(rew)

    for _ in range(20):
        one()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    s.synchronize()
    eager = (time.perf_counter() - t0) / steps
    ref_state = env.get_state().copy()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        one()
    for _ in range(20):
        graph.replay()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    s.synchronize()
    cap = (time.perf_counter() - t0) / steps
    st = env.get_state()
    steps_c, ticks_c = env.propagator.get_counters()
print("K %d n %d: eager %.2f us per step, graph replay %.2f us per step (x%.2f); ticks advanced to %d..%d; finite %s" % (K, n, eager * 1e6, cap * 1e6, eager / cap, ticks_c.min(), ticks_c.max(), bool(torch.isfinite(ret).all())))
env.close()
