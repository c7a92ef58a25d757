"""Multi-GPU sharding of independent environments, one process per GPU.

Spacecraft never interact (the reference owns exactly one ``scObject`` per simulator,
simulators/leoPowerAttitudeSimulator.py:213), so the step path has NO collective: rank ``r``
owns the contiguous env-index range ``shard_range(n_total, r, world)`` and steps it with its own
handle and stream.  The only exchange step is delivering the observation batch to whoever
consumes all of it: an all-gather (or gather to one root) of the per-rank ``f64[5][n_local]``
shards — RCCL over xGMI with ``backend="nccl"``, gloo on CPU in the tests.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous, balanced env-index range [lo, hi) of ``rank`` (first ``n_total % world`` ranks
    get one extra env)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


def local_obs_tensor(prop):
    """This rank's observations as a contiguous torch tensor (5, n_local): zero-copy view of the
    library's device buffer when the propagator exposes one, else a CPU tensor from host copies."""
    import torch

    if hasattr(prop, "device_views"):
        prop.sync()  # the step kernel runs on the handle's own stream
        v = prop.device_views()
        return torch.as_tensor(v["obs"], device="cuda").contiguous()
    return torch.from_numpy(np.ascontiguousarray(prop.get_obs()[0]))


def gather_observations(prop, dist, dst=None, group=None):
    """All-gather (``dst=None``) or gather-to-``dst`` of the observation shards.

    Equal shard sizes take the single-call ``all_gather_into_tensor`` path (one RCCL collective,
    every rank receives ``(world, 5, n_local)``); ragged shards fall back to ``all_gather`` on a
    padded buffer and are trimmed.  Returns the stacked tensor (or ``None`` on non-root ranks
    when ``dst`` is given)."""
    import torch

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local_obs_tensor(prop)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()   # gloo moves bytes through the host (CPU tests, single-GPU rehearsals)
    n_local = local.shape[1]
    sizes = [None] * world
    dist.all_gather_object(sizes, n_local, group=group)
    if len(set(sizes)) == 1:
        if dst is None:
            # concatenation along dim 0 is the layout both RCCL and gloo accept; view as (world, 5, n)
            out = torch.empty((world * local.shape[0], n_local), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local, group=group)
            return out.view(world, local.shape[0], n_local)
        bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, bufs, dst=dst, group=group)
        return torch.stack(bufs) if rank == dst else None
    n_max = max(sizes)
    padded = torch.zeros((5, n_max), dtype=local.dtype, device=local.device)
    padded[:, :n_local] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    out = [b[:, :s] for b, s in zip(bufs, sizes)]
    if dst is not None and rank != dst:
        return None
    return out


def concat_shards(gathered):
    """(world, 5, n_local) tensor or list of (5, n_r) tensors -> (5, n_total) in env-index order."""
    import torch

    if isinstance(gathered, (list, tuple)):
        return torch.cat(list(gathered), dim=1)
    return gathered.permute(1, 0, 2).reshape(gathered.shape[1], -1)
