"""Worker for tests/test_gpu_rccl.py: one rank on a real GPU, backend nccl (= RCCL on ROCm), HIP propagator.
The GPU box has one card, so the world is a single rank: what this exercises is the RCCL code path itself
(communicator bound to the device, the zero-copy device view handed to all_gather_into_tensor / gather,
the reward all-reduce) — the multi-rank bookkeeping is covered on CPU by tests/test_parallel_gloo.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    from basilisk_env_amd._lib import GRAV_PM_J2
    from basilisk_env_amd.parallel import concat_shards, gather_observations, shard_range
    from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
    from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

    n_total, out_dir = int(sys.argv[1]), sys.argv[2]
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(n_total, rank, world)
    cfg = default_config(4, GRAV_PM_J2)
    ic_all = sample_ic_batch(n_total, 4, seed=42)
    prop = BatchedPropagator(cfg, hi - lo, device=local)
    prop.reset(ic_all[:, lo:hi])
    actions = (np.arange(n_total) % 3).astype(np.int32)
    for k in (10, 7):
        prop.step(actions[lo:hi], k)
    gathered = gather_observations(prop, dist)
    assert gathered.is_cuda and gathered.shape == (world, 5, hi - lo)
    full = concat_shards(gathered)
    rooted = gather_observations(prop, dist, dst=0)
    rew = torch.tensor([prop.batch_stats()[0]], dtype=torch.float64, device="cuda")
    dist.all_reduce(rew)
    torch.cuda.synchronize()
    # the direct librccl leg in its rank-major form (one f64[6][n_r] block per rank), RCCL's own rank count, and the batch
    # scalars' all-reduce called TWICE between two steps (out of place: the second call must give the same sums)
    import ctypes
    from basilisk_env_amd import _hip
    from basilisk_env_amd.parallel import DirectRcclGather
    d = DirectRcclGather(prop, dist, root=0, rows=7, layout="rank-major")
    d.enqueue()
    sums = []
    for _ in range(2):
        p = d.all_reduce_stats()
        host = (ctypes.c_double * 2)()
        prop.sync()
        _hip.check(_hip.runtime().hipMemcpyAsync(ctypes.cast(host, ctypes.c_void_p), ctypes.c_void_p(p), 16, _hip.hipMemcpyDeviceToHost, ctypes.c_void_p(0)), "hipMemcpyAsync")
        torch.cuda.synchronize()
        sums.append([host[0], host[1]])
    rb = d.result_blocks()
    blk_obs = torch.cat([torch.as_tensor(b["obs"], device="cuda") for b in rb["blocks"]], dim=1)
    blk_rew = torch.cat([torch.as_tensor(b["reward"], device="cuda") for b in rb["blocks"]])
    blk_why = torch.as_tensor(rb["reason"], device="cuda")
    meta = {"comm_count": d.comm_count(), "messages_on_root": d.messages_on_root, "sums": sums}
    if rank == 0:
        np.save(os.path.join(out_dir, "obs_full.npy"), full.cpu().numpy())
        np.save(os.path.join(out_dir, "obs_root.npy"), concat_shards(rooted).cpu().numpy())
        np.save(os.path.join(out_dir, "rew_sum.npy"), rew.cpu().numpy())
        np.save(os.path.join(out_dir, "rm_obs.npy"), blk_obs.cpu().numpy())
        np.save(os.path.join(out_dir, "rm_rew.npy"), blk_rew.cpu().numpy())
        np.save(os.path.join(out_dir, "rm_why.npy"), blk_why.cpu().numpy())
        import json
        with open(os.path.join(out_dir, "meta.json"), "w") as f:
            json.dump(meta, f)
    dist.barrier()
    d.close()
    prop.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
