"""One batch of independent spacecraft over several GPUs of one node, driven by ONE process.

Spacecraft never interact (the reference owns exactly one ``scObject`` per simulator,
simulators/leoPowerAttitudeSimulator.py:213), so a batch is cut into contiguous env-index ranges
(``parallel.shard_range``), one propagator handle + HIP stream per device, and stepping needs NO collective:
``step`` enqueues one launch per device and returns, the devices run concurrently.  The results reach the host
through per-device 2-D async copies into ONE pinned buffer (each device writes its own columns of the
``[5][n_total]`` block, all DMA engines at once, one stream synchronisation per device afterwards), or stay on
the GPUs and are gathered to one device with the direct RCCL leg (``gather_obs_device``).

``ShardedPropagator`` has the ``BatchedPropagator`` interface, so everything above the propagator — the
``LeoPowerAttVecEnv`` host logic, episode statistics, auto-reset — is the single-GPU code unchanged, and a
sharded batch gives bit-identical results to the unsharded one (each handle knows the global index of its env 0,
``bsk_set_env_base``, for the device-side reset's slot rule).  ``ShardedVecEnv`` is the one-line composition a
stable-baselines process instantiates to put several GPUs behind one ``VecEnv``.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ._lib import n_fields
from .parallel import shard_range


class ShardedPropagator(object):
    def __init__(self, cfg, n_envs, devices=(0,), propagator_factory=None, device=None, stream=None):
        """``devices``: one entry per shard (a device may repeat: each entry gets its own handle and stream).
        ``propagator_factory(cfg, n, device=d)``: the per-shard engine (default BatchedPropagator)."""
        if propagator_factory is None:
            from .simulators.dynamics.propagator import BatchedPropagator as propagator_factory
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        if n_envs < len(self.devices):
            raise ValueError("fewer envs (%d) than shards (%d)" % (n_envs, len(self.devices)))
        self.cfg = cfg.copy()
        self.n_envs = int(n_envs)
        self.n_rw = int(cfg.n_rw)
        self.n_fields = n_fields(self.n_rw)
        self.device = self.devices[0]
        world = len(self.devices)
        self.ranges = [shard_range(self.n_envs, r, world) for r in range(world)]
        self.sizes = [hi - lo for lo, hi in self.ranges]
        self.shards = []
        self._host = None          # pinned [5 obs rows + reward][n] f64 and [n] u8, made on first get_obs
        self._comms = None
        self._stats_out = None     # per-device result blocks of all_reduce_stats_device (out of place)
        self._gather_buf = None
        self._pool_exec = None
        try:
            for (lo, hi), d in zip(self.ranges, self.devices):
                p = propagator_factory(cfg, hi - lo, device=d)
                p.set_env_base(lo)
                self.shards.append(p)
        except Exception:
            self.close()
            raise
        self._pool_exec = ThreadPoolExecutor(max_workers=world, thread_name_prefix="bsk-shard")

    # ------------------------------------------------------------------ plumbing
    def _each(self, fn):
        """fn(shard, lo, hi) on every shard, concurrently (blocking library calls release the GIL); results in shard order."""
        if len(self.shards) == 1:
            return [fn(self.shards[0], *self.ranges[0])]
        futs = [self._pool_exec.submit(fn, p, lo, hi) for p, (lo, hi) in zip(self.shards, self.ranges)]
        return [f.result() for f in futs]

    def close(self):
        for c in self._comms or []:
            c.destroy()
        self._comms = None
        if self._gather_buf is not None:
            self._gather_buf.free()
            self._gather_buf = None
        for b in self._stats_out or []:
            b.free()
        self._stats_out = None
        for p in self.shards:
            p.close()
        self.shards = []
        if self._host is not None:
            for b in self._host["bufs"]:
                b.free()
            self._host = None
        ex = getattr(self, "_pool_exec", None)
        if ex is not None:
            ex.shutdown(wait=True)
            self._pool_exec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def set_gravity_sh(self, degree, cbar, sbar):
        self._each(lambda p, lo, hi: p.set_gravity_sh(degree, cbar, sbar))

    def reset(self, ic, mask=None):
        ic = np.asarray(ic, dtype=np.float64)
        if ic.shape != (self.n_fields, self.n_envs):
            raise ValueError("ic must have shape (%d, %d), got %r" % (self.n_fields, self.n_envs, ic.shape))
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._each(lambda p, lo, hi: p.reset(np.ascontiguousarray(ic[:, lo:hi]), None if m is None else m[lo:hi]))

    def get_state(self):
        return np.concatenate(self._each(lambda p, lo, hi: p.get_state()), axis=1)

    def set_state(self, state):
        state = np.asarray(state, dtype=np.float64)
        self._each(lambda p, lo, hi: p.set_state(np.ascontiguousarray(state[:, lo:hi])))

    def get_counters(self):
        res = self._each(lambda p, lo, hi: p.get_counters())
        return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])

    def set_counters(self, steps, ticks):
        steps, ticks = np.asarray(steps, np.int32), np.asarray(ticks, np.int32)
        self._each(lambda p, lo, hi: p.set_counters(steps[lo:hi], ticks[lo:hi]))

    def set_sim_time(self, t):
        for p in self.shards:
            p.set_sim_time(t)

    def set_env_base(self, base):
        for p, (lo, _) in zip(self.shards, self.ranges):
            p.set_env_base(int(base) + lo)

    # ------------------------------------------------------------------ stepping (no collective)
    def step(self, actions, substeps):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        self._last_actions = a   # the shards' async H2D copies read slices of this array
        for p, (lo, hi) in zip(self.shards, self.ranges):
            p.step(a[lo:hi], substeps)      # asynchronous: H2D of the slice + one launch on the shard's stream

    def step_device(self, d_action_ptrs, substeps):
        """Actions already resident on each device: one device pointer per shard."""
        for p, ptr in zip(self.shards, d_action_ptrs):
            p.step_device(ptr, substeps)

    def _pinned(self):
        if self._host is None:
            from . import _hip
            n = self.n_envs
            f64 = _hip.PinnedBuffer(6 * n * 8)
            u8 = _hip.PinnedBuffer(n)
            blk = f64.array.view(np.float64).reshape(6, n)
            self._host = {"bufs": [f64, u8], "obs": blk[:5], "rew": blk[5], "why": u8.array, "ptr_f64": f64.ptr, "ptr_u8": u8.ptr}
        return self._host

    def get_obs(self):
        """-> obs (5, N), reward (N,), done (N,) bool, reason (N,) uint8.  Device shards with zero-copy views deliver
        straight into one pinned block (every device's copies in flight at once); other engines (the tests'
        oracle stand-in) are read shard by shard."""
        if not all(hasattr(p, "device_views") and hasattr(p, "stream_ptr") for p in self.shards):
            res = self._each(lambda p, lo, hi: p.get_obs())
            return (np.concatenate([r[0] for r in res], axis=1), np.concatenate([r[1] for r in res]),
                    np.concatenate([r[2] for r in res]), np.concatenate([r[3] for r in res]))
        from . import _hip
        h = self._pinned()
        n = self.n_envs
        streams = []
        for p, (lo, hi) in zip(self.shards, self.ranges):
            v, st = p.device_views(), p.stream_ptr()
            pitch = v["stride"] * 8
            w = (hi - lo) * 8
            with _hip.device_guard(p.device):      # (the calling thread's current device is the caller's business: restored)
                _hip.memcpy2d_async(h["ptr_f64"] + lo * 8, n * 8, v["obs"].__cuda_array_interface__["data"][0], pitch, w, 5,
                                    _hip.hipMemcpyDeviceToHost, st)
                _hip.memcpy2d_async(h["ptr_f64"] + (5 * n + lo) * 8, n * 8, v["reward"].__cuda_array_interface__["data"][0], w, w, 1,
                                    _hip.hipMemcpyDeviceToHost, st)
                _hip.memcpy2d_async(h["ptr_u8"] + lo, n, v["reason"].__cuda_array_interface__["data"][0], hi - lo, hi - lo, 1,
                                    _hip.hipMemcpyDeviceToHost, st)
            streams.append(st)
        for st in streams:
            _hip.stream_sync(st)
        why = h["why"].copy()
        return h["obs"].copy(), h["rew"].copy(), why != 0, why

    def batch_stats(self):
        res = self._each(lambda p, lo, hi: p.batch_stats())
        return float(sum(r[0] for r in res)), int(sum(r[1] for r in res))

    def sync(self):
        for p in self.shards:
            p.sync()

    # ------------------------------------------------------------------ device-side reset
    def set_ic_pool(self, ic_pool):
        self._each(lambda p, lo, hi: p.set_ic_pool(ic_pool))

    def sample_ic_pool(self, n_pool, seed):
        self._each(lambda p, lo, hi: p.sample_ic_pool(n_pool, seed))   # same key on every device: the same pool

    def get_ic_pool(self):
        return self.shards[0].get_ic_pool()

    def reset_from_pool(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._each(lambda p, lo, hi: p.reset_from_pool(None if m is None else m[lo:hi]))

    def get_terminal_obs(self):
        res = self._each(lambda p, lo, hi: p.get_terminal_obs())
        return np.concatenate([r[0] for r in res], axis=1), np.concatenate([r[1] for r in res])

    # ------------------------------------------------------------------ measurement / hand-off
    def kernel_info(self):
        return self.shards[0].kernel_info()

    def device_views(self):
        """Per shard: (lo, hi, device, views)."""
        return [(lo, hi, p.device, p.device_views()) for p, (lo, hi) in zip(self.shards, self.ranges)]

    def gather_step_outputs_device(self, root=0):
        """What a trainer on ONE GPU needs of the whole batch - observations ``f64[5][n_total]``, rewards ``f64[n_total]``,
        done reasons ``u8[n_total]`` (done = reason != 0) - on the root shard's GPU without touching the host: ONE group of
        ncclSend / ncclRecv (seven rows per rank) from every other device's buffers straight into the root's (rccl.py),
        enqueued on the handles' own streams.  Returns zero-copy device views ``{"obs", "reward", "reason"}`` (valid until the
        next gather); the caller orders its consumer after the root shard's stream (``shards[root].stream_ptr()``).
        SURVEY.md section 8(e)."""
        from . import _hip, rccl
        from .simulators.dynamics.propagator import _DevArray
        world, n = len(self.shards), self.n_envs
        if self._gather_buf is None:
            self._gather_buf = _hip.DeviceBuffer(6 * n * 8 + n, self.shards[root].device)       # obs | reward | reason
            self._gather_root = root
        elif self._gather_root != root:
            raise ValueError("gather buffer lives on shard %d's device" % self._gather_root)
        if world > 1 and self._comms is None:
            self._comms = rccl.Comm.init_all(self.devices)
        out = self._gather_buf.ptr
        out_obs, out_rew, out_why = out, out + 5 * n * 8, out + 6 * n * 8
        bufs, streams = [], []
        for p in self.shards:
            v = p.device_views()
            ptr = lambda k: v[k].__cuda_array_interface__["data"][0]
            bufs.append(rccl.step_output_bufs(ptr("obs"), v["stride"] * 8, ptr("reward"), ptr("reason"), out_obs, out_rew, out_why))
            streams.append(p.stream_ptr())
        if world > 1:
            rccl.group_start()
            for r in range(world):
                rccl.enqueue_gather(self._comms[r], streams[r], root, self.sizes, bufs[r])
            rccl.group_end()
        rroot = self._comms[root] if self._comms else rccl.Comm(None, root, world, self.shards[root].device)
        rccl.copy_own(rroot, streams[root], root, self.sizes, bufs[root])
        kw = {"owner": self.shards[root], "device": self.shards[root].device, "stream": streams[root]}
        return {"obs": _DevArray(out_obs, (5, n), "<f8", **kw), "reward": _DevArray(out_rew, (n,), "<f8", **kw),
                "reason": _DevArray(out_why, (n,), "|u1", **kw), "bytes_over_fabric": rccl.gather_bytes(self.sizes, bufs[root], root)}

    def gather_obs_device(self, root=0):
        """The observation part of gather_step_outputs_device (rewards and reasons travel in the same group)."""
        return self.gather_step_outputs_device(root)["obs"]

    def rollout(self, n_steps, substeps, actions=None, constant_action=0):
        """``BatchedPropagator.rollout`` on every shard (one thread per device), histories joined in env order."""
        acts = None if actions is None else np.ascontiguousarray(actions, dtype=np.int32)
        res = self._each(lambda p, lo, hi: p.rollout(n_steps, substeps, None if acts is None else acts[:, lo:hi], constant_action))
        return (np.concatenate([r[0] for r in res], axis=2), np.concatenate([r[1] for r in res], axis=1),
                np.concatenate([r[2] for r in res], axis=1))

    def set_step_stats(self, on=True):
        """Every shard's step launches form the per-wave reward sums themselves (bsk_set_step_stats): for loops that call
        all_reduce_stats_device / batch_stats after every step."""
        for p in self.shards:
            p.set_step_stats(on)

    def all_reduce_stats_device(self):
        """Batch scalars of the last step on EVERY shard's GPU: one ncclAllReduce of two doubles {sum of rewards, number of done
        envs} on the handles' streams, operands produced on the device (bsk_get_batch_stats_device) - no host value involved.
        -> list of device pointers (f64[2]), one per shard; a single shard needs no collective."""
        from . import _hip, rccl
        ptrs = [p.batch_stats_device() for p in self.shards]
        if len(self.shards) > 1:
            if self._comms is None:
                self._comms = rccl.Comm.init_all(self.devices)
            if self._stats_out is None:
                self._stats_out = [_hip.DeviceBuffer(16, d) for d in self.devices]
            # OUT OF PLACE: the handles' own blocks are only refreshed after the next step - reduced in place, a second call
            # between two steps would add the already-summed values once more per shard
            rccl.group_start()
            for c, p, ptr, out in zip(self._comms, self.shards, ptrs, self._stats_out):
                rccl.all_reduce_sum_f64(c, p.stream_ptr(), ptr, out.ptr, 2)
            rccl.group_end()
            return [o.ptr for o in self._stats_out]
        return ptrs


def ShardedVecEnv(num_envs, devices=(0,), propagator_factory=None, **kw):
    """``LeoPowerAttVecEnv`` over several GPUs of one node in one process: the same class, its propagator a
    ``ShardedPropagator`` (one handle + stream per entry of ``devices``, env-index ranges, no step-path collective).
    Every keyword of ``LeoPowerAttVecEnv`` applies."""
    from .envs.leoPowerAttitudeVecEnv import LeoPowerAttVecEnv
    devices = list(devices)

    def factory(cfg, n, device=0, **_):
        return ShardedPropagator(cfg, n, devices=devices, propagator_factory=propagator_factory)

    return LeoPowerAttVecEnv(num_envs, device=devices[0], propagator_factory=factory, **kw)
