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


def _order_after_handle(prop, torch):
    """Make torch's current stream wait for the work queued on the propagator's own stream (the step kernel) with a
    device-side dependency — an event the GPU waits on — instead of blocking the host on the stream."""
    if hasattr(prop, "stream_ptr"):
        dev = torch.device("cuda", getattr(prop, "device", torch.cuda.current_device()))
        cur = torch.cuda.current_stream(dev)
        hs = prop.stream_ptr()
        if hs and cur.cuda_stream != hs:
            cur.wait_stream(torch.cuda.ExternalStream(hs, device=dev))
    else:
        prop.sync()


def local_obs_tensor(prop):
    """This rank's observations as a contiguous torch tensor (5, n_local): zero-copy view of the
    library's device buffer when the propagator exposes one, else a CPU tensor from host copies."""
    import torch

    if hasattr(prop, "device_views"):
        _order_after_handle(prop, torch)   # the step kernel runs on the handle's own stream
        v = prop.device_views()
        return torch.as_tensor(v["obs"], device="cuda").contiguous()
    return torch.from_numpy(np.ascontiguousarray(prop.get_obs()[0]))


def release_to_handle(prop):
    """The other half of the ordering: make the handle's stream wait for what torch's current stream has queued so far (the
    copy of the padded rows, the collective that reads the zero-copy alias), so that the NEXT step kernel cannot overwrite
    the observation buffer while it is still being read.  Device-side dependency, no host synchronisation."""
    import torch

    if hasattr(prop, "stream_ptr"):
        dev = torch.device("cuda", getattr(prop, "device", torch.cuda.current_device()))
        cur = torch.cuda.current_stream(dev)
        hs = prop.stream_ptr()
        if hs and cur.cuda_stream != hs:
            torch.cuda.ExternalStream(hs, device=dev).wait_stream(cur)


class ObsGatherer(object):
    """The exchange step, set up ONCE per (propagator, group): shard sizes are static (env-index ranges), so they are
    exchanged here and never again; ``gather`` / ``all_gather`` then run exactly one collective per call."""

    def __init__(self, prop, dist, group=None):
        self.prop, self.dist, self.group = prop, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.cpu = dist.get_backend(group) == "gloo"
        n_local = int(prop.n_envs)
        sizes = [None] * self.world
        dist.all_gather_object(sizes, n_local, group=group)     # pickled host collective: construction only
        self.sizes = [int(x) for x in sizes]
        self.equal = len(set(self.sizes)) == 1
        self.n_local = n_local

    def _release(self, local):
        """Contract: a gather reads the library's observation buffer (or a copy queued on torch's stream); the handle's
        stream is made to wait for those reads before its next step kernel may run (release_to_handle)."""
        if local.is_cuda:
            release_to_handle(self.prop)

    def _local(self):
        local = local_obs_tensor(self.prop)
        if local.is_cuda and self.cpu:
            local = local.cpu()   # gloo moves bytes through the host (CPU tests, single-GPU rehearsals)
        return local

    def all_gather(self):
        import torch
        dist, local = self.dist, self._local()
        if self.equal:
            # concatenation along dim 0 is the layout both RCCL and gloo accept; view as (world, 5, n)
            out = torch.empty((self.world * local.shape[0], self.n_local), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local, group=self.group)
            self._release(local)
            return out.view(self.world, local.shape[0], self.n_local)
        n_max = max(self.sizes)
        padded = torch.zeros((5, n_max), dtype=local.dtype, device=local.device)
        padded[:, :self.n_local] = local
        bufs = [torch.empty_like(padded) for _ in range(self.world)]
        dist.all_gather(bufs, padded, group=self.group)
        self._release(local)
        return [b[:, :s] for b, s in zip(bufs, self.sizes)]

    def gather(self, dst=0):
        import torch
        dist, local = self.dist, self._local()
        if self.equal:
            bufs = [torch.empty_like(local) for _ in range(self.world)] if self.rank == dst else None
            dist.gather(local, bufs, dst=dst, group=self.group)
            self._release(local)
            return torch.stack(bufs) if self.rank == dst else None
        out = self.all_gather()
        return out if self.rank == dst else None


def gather_observations(prop, dist, dst=None, group=None):
    """All-gather (``dst=None``) or gather-to-``dst`` of the observation shards.

    Equal shard sizes take the single-call ``all_gather_into_tensor`` path (one RCCL collective,
    every rank receives ``(world, 5, n_local)``); ragged shards fall back to ``all_gather`` on a
    padded buffer and are trimmed.  Returns the stacked tensor (or ``None`` on non-root ranks
    when ``dst`` is given).  The shard sizes are exchanged on the first call for a (propagator, group) and cached
    on the propagator (``ObsGatherer``)."""
    cache = prop.__dict__.setdefault("_obs_gatherers", {})
    key = (id(dist), id(group))
    g = cache.get(key)
    if g is None:
        g = cache[key] = ObsGatherer(prop, dist, group)
    return g.all_gather() if dst is None else g.gather(dst)


class DirectRcclGather(object):
    """One process per GPU, the direct leg: this rank's communicator comes from ``rccl.Comm.init_rank`` (unique id
    broadcast once through ``dist``), the gather itself is ONE group of ncclSend / ncclRecv on the propagator handle's OWN
    stream from the library's buffers into the root's device buffers - observations ``[5][n_total]``, and with
    ``rows=7`` (default) rewards ``[n_total]`` and done reasons ``u8[n_total]`` as well (what a
    consumer on the root GPU needs to train on) - no torch tensor, no staging copy, no host synchronisation between the
    step kernel and the exchange.  ``all_reduce_stats`` is the other collective of the path: two doubles per rank."""

    def __init__(self, prop, dist, root=0, group=None, rows=7, layout="columns"):
        """``layout``: "columns" - the root's buffers are ``[5][n_total]`` / ``[n_total]`` / ``u8[n_total]`` in env-index order, one
        message per row and rank (seven per rank); "rank-major" - the root's buffer holds one ``f64[6][n_r]`` block per rank
        (observation rows, then the reward row) and the reasons in rank order: ONE f64 message + one u8 message per rank whose
        shard is contiguous (rccl.py: rank_major_contiguous), fourteen receives on the root of an 8-GPU node instead of 49."""
        from . import _hip, rccl
        if rows not in (5, 7):
            raise ValueError("rows: 5 (observations) or 7 (+ reward, reason)")
        if layout not in ("columns", "rank-major") or (layout == "rank-major" and rows != 7):
            raise ValueError("layout: 'columns' or 'rank-major' (the latter always moves the seven rows)")
        self.prop, self.root, self.rows, self.layout = prop, int(root), int(rows), layout
        self._stats_out = None
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        box = [rccl.unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        sizes = [None] * self.world
        dist.all_gather_object(sizes, int(prop.n_envs), group=group)
        self.sizes = [int(x) for x in sizes]
        n = self.n_total = sum(self.sizes)
        self.comm = rccl.Comm.init_rank(self.world, self.rank, box[0], prop.device)
        self.out = _hip.DeviceBuffer(6 * n * 8 + n, prop.device) if self.rank == self.root else None     # obs | reward | reason
        o = self.out.ptr if self.out is not None else 0
        v = prop.device_views()
        ptr = lambda k: v[k].__cuda_array_interface__["data"][0]
        self.bufs = rccl.step_output_bufs(ptr("obs"), v["stride"] * 8, ptr("reward"), ptr("reason"), o, o + 5 * n * 8, o + 6 * n * 8)
        if self.rows == 5:
            self.bufs = self.bufs[:1]
        self.bytes_over_fabric = rccl.gather_bytes(self.sizes, self.bufs, self.root)
        others = [s for r, s in enumerate(self.sizes) if r != self.root and s]
        if layout == "rank-major":
            self.rm = rccl.RankMajorBufs(ptr("obs"), v["stride"] * 8, ptr("reward"), ptr("reason"), o, o + 6 * n * 8)
            # one block or six rows: each rank says what ITS buffers are (real pitch, reward directly behind the observation rows),
            # exchanged here once - the root posts the receives that match, whatever row pitch a rank's library was built with
            flags = [None] * self.world
            dist.all_gather_object(flags, bool(self.rm.contiguous(prop.n_envs)), group=group)
            self.contig = [bool(f) for f in flags]
            self.messages_on_root = rccl.rank_major_messages(self.sizes, self.root, self.contig)
        else:
            self.messages_on_root = len(others) * sum(b.rows for b in self.bufs)
        self.stream = prop.stream_ptr()

    def comm_count(self):
        """Ranks the communicator spans as RCCL itself reports them (ncclCommCount)."""
        from . import rccl
        return rccl.comm_count(self.comm)

    def enqueue(self):
        """Queue one gather behind whatever the handle's stream holds (asynchronous)."""
        from . import rccl
        rccl.group_start()
        if self.layout == "rank-major":
            rccl.enqueue_gather_rank_major(self.comm, self.stream, self.root, self.sizes, self.rm, self.contig)
        else:
            rccl.enqueue_gather(self.comm, self.stream, self.root, self.sizes, self.bufs)
        rccl.group_end()
        if self.layout == "rank-major":
            rccl.copy_own_rank_major(self.comm, self.stream, self.root, self.sizes, self.rm)
        else:
            rccl.copy_own(self.comm, self.stream, self.root, self.sizes, self.bufs)

    def all_reduce_stats(self):
        """{sum of rewards, number of done envs} of the whole batch on every rank's GPU: the device-side partial of this rank
        (bsk_get_batch_stats_device; untouched by the collective) all-reduced OUT OF PLACE on the handle's stream into a
        result block of this object's own - calling it again before the next step gives the same sums again.  A loop that calls it
        after EVERY step switches prop.set_step_stats(True) on first: the step launch then forms the per-wave sums itself and the
        operand costs one small launch (the join) instead of two.
        -> device pointer of the result f64[2]."""
        from . import _hip, rccl
        if self._stats_out is None:
            self._stats_out = _hip.DeviceBuffer(16, self.prop.device)
        rccl.all_reduce_sum_f64(self.comm, self.stream, self.prop.batch_stats_device(), self._stats_out.ptr, 2)
        return self._stats_out.ptr

    def _view(self, k, shape, typestr):
        from .simulators.dynamics.propagator import _DevArray
        if self.out is None:
            return None
        return _DevArray(self.bufs[k].out_ptr, shape, typestr, owner=self.prop, device=self.prop.device, stream=self.stream)

    def result_view(self):
        """Root only: zero-copy view (5, n_total) of the gathered observations, env-index order."""
        return self._view(0, (5, self.n_total), "<f8")

    def result_views(self):
        """Root only: {"obs" (5, n_total), "reward" (n_total,), "reason" (n_total,) uint8} - layout "columns"."""
        if self.layout == "rank-major":
            raise ValueError("rank-major layout: result_blocks()")
        if self.out is None or self.rows != 7:
            return None if self.out is None else {"obs": self.result_view()}
        return {"obs": self.result_view(), "reward": self._view(1, (self.n_total,), "<f8"), "reason": self._view(2, (self.n_total,), "|u1")}

    def result_blocks(self):
        """Root only, layout "rank-major": per rank {"lo", "hi" (env-index range), "obs" (5, n_r), "reward" (n_r,)} zero-copy views
        of its block, and "reason" (n_total,) uint8 in env-index order for the whole batch."""
        from . import rccl
        from .simulators.dynamics.propagator import _DevArray
        if self.out is None:
            return None
        offs, n = rccl.column_offsets(self.sizes)
        kw = {"owner": self.prop, "device": self.prop.device, "stream": self.stream}
        blocks = []
        for r, n_r in enumerate(self.sizes):
            base = self.rm.out_f64 + 48 * offs[r]
            blocks.append({"lo": offs[r], "hi": offs[r] + n_r, "obs": _DevArray(base, (5, n_r), "<f8", **kw),
                           "reward": _DevArray(base + 40 * n_r, (n_r,), "<f8", **kw)})
        return {"blocks": blocks, "reason": _DevArray(self.rm.out_u8, (n,), "|u1", **kw)}

    def close(self):
        self.comm.destroy()
        if self.out is not None:
            self.out.free()
            self.out = None
        if self._stats_out is not None:
            self._stats_out.free()
            self._stats_out = None


def concat_shards(gathered):
    """(world, 5, n_local) tensor or list of (5, n_r) tensors -> (5, n_total) in env-index order."""
    import torch

    if isinstance(gathered, (list, tuple)):
        return torch.cat(list(gathered), dim=1)
    return gathered.permute(1, 0, 2).reshape(gathered.shape[1], -1)
