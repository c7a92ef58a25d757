"""The device-resident RL loop captured in a HIP graph: U iterations (policy matmul + argmax + step kernel with device-side
auto-reset) per graph, replayed.  Checks the graph loop against the eager loop (same seeds -> identical episode returns), then
times both.   usage: tools/exp/rl_graph.py [envs] [substeps] [iterations per graph] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basilisk_env_amd.envs import LeoPowerAttVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
U = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
steps -= steps % U


def make(side):
    env = LeoPowerAttVecEnv(n, n_rw=4, step_duration=0.1 * K, seed=0, device_reset_pool=4096, device_sampler=True, stream=side.cuda_stream)
    ob = env.reset_tensors()
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(5, 3, dtype=torch.float64, device="cuda", generator=g)
    return env, ob, w


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    # eager
    env, ob, w = make(side)
    for _ in range(U * 2):
        ob, _, _, info = env.step_tensors((ob.reshape(n, 5) @ w).argmax(dim=1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ob, _, _, info = env.step_tensors((ob.reshape(n, 5) @ w).argmax(dim=1))
    torch.cuda.synchronize()
    el_eager = time.perf_counter() - t0
    ret_eager = info["episode_return"].clone(); eps_eager = info["episodes"].clone()
    env.close()

    # graph: the env's output buffers are fixed device addresses, the policy writes its actions into a fixed tensor
    env, ob, w = make(side)
    act = torch.zeros(n, dtype=torch.int64, device="cuda")
    logits = torch.zeros(n, 3, dtype=torch.float64, device="cuda")

    def one():
        torch.matmul(env._torch_views()["obs_n51"].reshape(n, 5), w, out=logits)
        torch.argmax(logits, dim=1, out=act)
        return env.step_tensors(act)

    for _ in range(U * 2):                     # warm-up on the capture stream (same count as the eager loop's)
        one()
    torch.cuda.synchronize()
    c0 = env.propagator.debug_counters()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(U):
            _, _, _, info = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // U):
        graph.replay()
    torch.cuda.synchronize()
    el_graph = time.perf_counter() - t0
    c1 = env.propagator.debug_counters()
    ret_graph = info["episode_return"].clone(); eps_graph = info["episodes"].clone()
    same = bool(torch.equal(ret_eager, ret_graph)) and bool(torch.equal(eps_eager, eps_graph))
    env.close()
print("envs %d K %d: eager %.2f us/step, graph (%d iterations per graph) %.2f us/step; identical episode returns and counts: %s; "
      "copies/syncs issued by the library during capture + replay: %s"
      % (n, K, el_eager / steps * 1e6, U, el_graph / steps * 1e6, same, tuple(b - a for a, b in zip(c0, c1))))
sys.exit(0 if same else 1)
