"""BatchedPropagator — Python face of the HIP propagator (libbskgpu.so, include/bskgpu.h).

Replaces, for N spacecraft at once, what the reference does per env step by calling into the
Basilisk engine: ``ConfigureStopTime`` + ``ExecuteSimulation`` + ``pullMultiMessageLogData``
(reference simulators/leoPowerAttitudeSimulator.py:590-619).  numpy arrays in, numpy arrays out;
device memory belongs to the library.  There is no CPU path: construction raises
``BskGpuUnavailable`` when the library or a gfx950 device is missing.
"""
import ctypes as C

import numpy as np

from ... import _lib
from ..._lib import FLAG_OBS_ROWMAJOR, BskConfig, check, n_fields


def pack_ic(n_rw, rN, vN, sigma, omega, wheelSpeeds=None, lext=None, charge=None, ucmd=None):
    """Assemble the SoA initial-condition block ``[n_fields, N]`` (field order of include/bskgpu.h).

    rN, vN, sigma, omega: (N,3); wheelSpeeds: (N,n_rw) [rad/s]; lext: (N,3) [N m]; charge: (N,) [W s].
    """
    rN = np.atleast_2d(np.asarray(rN, dtype=np.float64))
    n = rN.shape[0]
    nf = n_fields(n_rw)
    ic = np.zeros((nf, n), dtype=np.float64)
    ic[_lib.F_R:_lib.F_R + 3] = rN.T
    ic[_lib.F_V:_lib.F_V + 3] = np.atleast_2d(vN).T
    ic[_lib.F_SIGMA:_lib.F_SIGMA + 3] = np.atleast_2d(sigma).T
    ic[_lib.F_OMEGA:_lib.F_OMEGA + 3] = np.atleast_2d(omega).T
    if n_rw:
        ic[_lib.NF_BASE:_lib.NF_BASE + n_rw] = np.atleast_2d(wheelSpeeds).T
    t = _lib.NF_BASE + n_rw
    if lext is not None:
        ic[t + _lib.T_LEXT:t + _lib.T_LEXT + 3] = np.atleast_2d(lext).T
    if ucmd is not None:
        ic[t + _lib.T_UCMD:t + _lib.T_UCMD + n_rw] = np.atleast_2d(ucmd).T
    if charge is not None:
        ic[t + _lib.T_CHARGE] = np.asarray(charge, dtype=np.float64).reshape(n)
    return ic


class _DevArray(object):
    """Zero-copy view of a library-owned device buffer: ``__cuda_array_interface__`` (torch.as_tensor, cupy, numba)
    and DLPack (``torch.from_dlpack(view)``; device type kDLROCM).  The buffer belongs to the propagator, which the
    view keeps alive; its contents are valid once the work queued on the handle's stream has been ordered before
    the consumer's (same stream, a stream wait, or ``propagator.sync()``)."""

    def __init__(self, ptr, shape, typestr, strides=None, owner=None, device=0, stream=None):
        self._owner = owner
        self._device = int(device)
        self._stream = stream
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": strides,
        }

    def __dlpack_device__(self):
        from ..._dlpack import kDLROCM
        return (kDLROCM, self._device)

    def __dlpack__(self, stream=None, max_version=None, dl_device=None):
        """``stream``: the consumer's stream as the DLPack protocol passes it.  When it is the handle's own stream
        (a propagator created on the consumer's stream) nothing needs ordering; otherwise the handle's stream is
        drained first, so that the consumer never reads a buffer a queued step kernel is still writing."""
        from ..._dlpack import make_capsule
        if self._owner is not None and hasattr(self._owner, "sync"):
            same = stream is not None and self._stream is not None and int(stream) not in (0, 1, 2, -1) and int(stream) == int(self._stream)
            if not same:
                self._owner.sync()
        a = self.__cuda_array_interface__
        return make_capsule(a["data"][0], a["shape"], a["typestr"], a["strides"], device_id=self._device, owner=self)


class BatchedPropagator(object):
    def __init__(self, cfg, n_envs, device=0, stream=None):
        if not isinstance(cfg, BskConfig):
            raise TypeError("cfg must be a BskConfig")
        self._lib = _lib.load()
        self.cfg = cfg.copy()
        self.n_envs = int(n_envs)
        self.n_rw = int(cfg.n_rw)
        self.n_fields = n_fields(self.n_rw)
        self.device = int(device)
        h = C.c_void_p()
        check(self._lib.bsk_create(C.byref(self.cfg), self.n_envs, self.device,
                                   C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsk_destroy(self._h)
            self._h = None
        for name in ("_pin", "_pin_rm"):
            pin = getattr(self, name, None)
            if pin is not None:
                for b in pin["bufs"]:
                    b.free()
                setattr(self, name, None)

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():      # interpreter teardown: the HIP runtime may be gone already, the OS reclaims the rest
                return
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise RuntimeError("propagator is closed")
        return self._h

    def set_gravity_sh(self, degree, cbar, sbar):
        """Normalised spherical-harmonic coefficients, packed l*(l+1)/2+m (gravity_sh.sh_index);
        required before stepping a handle created with gravity_model = GRAV_SH."""
        from .gravity_sh import sh_size
        cbar = np.ascontiguousarray(cbar, dtype=np.float64)
        sbar = np.ascontiguousarray(sbar, dtype=np.float64)
        if cbar.shape != (sh_size(degree),) or sbar.shape != cbar.shape:
            raise ValueError("cbar/sbar must have %d entries for degree %d" % (sh_size(degree), degree))
        check(self._lib.bsk_set_gravity_sh(self._handle(), int(degree), cbar.ctypes.data, sbar.ctypes.data))

    # ------------------------------------------------------------------ state
    def reset(self, ic, mask=None):
        ic = np.ascontiguousarray(ic, dtype=np.float64)
        if ic.shape != (self.n_fields, self.n_envs):
            raise ValueError("ic must have shape (%d, %d), got %r" % (self.n_fields, self.n_envs, ic.shape))
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            if mask.shape != (self.n_envs,):
                raise ValueError("mask must have shape (%d,)" % self.n_envs)
            mp = mask.ctypes.data
        check(self._lib.bsk_reset(self._handle(), mp, ic.ctypes.data))

    def get_state(self):
        out = np.empty((self.n_fields, self.n_envs), dtype=np.float64)
        check(self._lib.bsk_get_state(self._handle(), out.ctypes.data))
        return out

    def set_state(self, state):
        state = np.ascontiguousarray(state, dtype=np.float64)
        if state.shape != (self.n_fields, self.n_envs):
            raise ValueError("state must have shape (%d, %d)" % (self.n_fields, self.n_envs))
        check(self._lib.bsk_set_state(self._handle(), state.ctypes.data))

    def get_counters(self):
        steps = np.empty(self.n_envs, dtype=np.int32)
        ticks = np.empty(self.n_envs, dtype=np.int32)
        check(self._lib.bsk_get_counters(self._handle(), steps.ctypes.data, ticks.ctypes.data))
        return steps, ticks

    def set_counters(self, steps, ticks):
        """Restore the per-env counters (with ``set_state``: a full checkpoint of the batch)."""
        steps = np.ascontiguousarray(steps, dtype=np.int32)
        ticks = np.ascontiguousarray(ticks, dtype=np.int32)
        if steps.shape != (self.n_envs,) or ticks.shape != (self.n_envs,):
            raise ValueError("steps and ticks must have shape (%d,)" % self.n_envs)
        check(self._lib.bsk_set_counters(self._handle(), steps.ctypes.data, ticks.ctypes.data))

    # ------------------------------------------------------------------ stepping
    def step(self, actions, substeps):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.shape != (self.n_envs,):
            raise ValueError("actions must have shape (%d,)" % self.n_envs)
        self._last_actions = a  # keep alive until the async H2D copy has been consumed
        check(self._lib.bsk_step(self._handle(), a.ctypes.data, int(substeps)))

    def step_device(self, d_actions_ptr, substeps, int64=False):
        """Actions already on the device: int32[n] (``int64=False``) or int64[n] (torch's argmax output, read in place)."""
        fn = self._lib.bsk_step_device_i64 if int64 else self._lib.bsk_step_device
        check(fn(self._handle(), C.c_void_p(int(d_actions_ptr)), int(substeps)))

    def step_n(self, n_steps, substeps, d_actions_ptr=None, constant_action=0, d_obs_hist=None, d_reward_hist=None, d_reason_hist=None):
        """Open-loop rollout: ``n_steps`` env steps in ONE launch (``bsk_step_n``; what the reference's mains do with a constant
        action, envs/leoPowerAttitudeEnvironment.py:218-231).  ``d_actions_ptr``: device pointer of int32[n_steps][n_envs], or None
        for ``constant_action``; the history arguments are device pointers (or None) of f64[n_steps][5][n_envs],
        f64[n_steps][n_envs], u8[n_steps][n_envs].  Asynchronous; the handle's buffers end as after ``n_steps`` single steps."""
        vp = lambda p: C.c_void_p(int(p)) if p else None      # noqa: E731
        check(self._lib.bsk_step_n(self._handle(), vp(d_actions_ptr), int(constant_action), int(substeps), int(n_steps),
                                   vp(d_obs_hist), vp(d_reward_hist), vp(d_reason_hist)))

    def rollout(self, n_steps, substeps, actions=None, constant_action=0):
        """``step_n`` with host arrays: ``actions`` int32 (n_steps, n_envs) or None -> (obs (n_steps, 5, n_envs), reward
        (n_steps, n_envs), reason (n_steps, n_envs) uint8) as numpy arrays.  Allocates device scratch per call and synchronises:
        the convenience form (tests, scripted evaluations); a training process hands ``step_n`` its own device buffers."""
        from ... import _hip
        n, T = self.n_envs, int(n_steps)
        rt = _hip.runtime()
        bufs = [_hip.DeviceBuffer(T * 5 * n * 8, self.device), _hip.DeviceBuffer(T * n * 8, self.device), _hip.DeviceBuffer(T * n, self.device)]
        act = None
        stream = C.c_void_p(self.stream_ptr())
        try:
            if actions is not None:
                a = np.ascontiguousarray(actions, dtype=np.int32)
                if a.shape != (T, n):
                    raise ValueError("actions must have shape (%d, %d)" % (T, n))
                act = _hip.DeviceBuffer(T * n * 4, self.device)
                _hip.check(rt.hipMemcpyAsync(C.c_void_p(act.ptr), C.c_void_p(a.ctypes.data), T * n * 4, _hip.hipMemcpyHostToDevice, stream), "hipMemcpyAsync")
                self.sync()                      # (pageable source: the copy must have left `a` before it goes out of scope)
            self.step_n(T, substeps, act.ptr if act else None, constant_action, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
            obs, rew, why = np.empty((T, 5, n)), np.empty((T, n)), np.empty((T, n), dtype=np.uint8)
            for dst, b in ((obs, bufs[0]), (rew, bufs[1]), (why, bufs[2])):
                _hip.check(rt.hipMemcpyAsync(C.c_void_p(dst.ctypes.data), C.c_void_p(b.ptr), dst.nbytes, _hip.hipMemcpyDeviceToHost, stream), "hipMemcpyAsync")
            self.sync()
        finally:
            for b in bufs + ([act] if act else []):
                b.free()
        return obs, rew, why

    pinned_read_back = True     # get_obs(copy=False) exists

    def get_obs(self, copy=True):
        """-> obs (5, N) f64, reward (N,) f64, done (N,) bool, reason (N,) uint8.

        ``copy=False``: obs / reward / reason are views of page-locked host buffers the propagator owns and the NEXT
        call overwrites - the device-to-host copies then run as plain DMA (no staging through a bounce buffer, no
        first-touch page faults on fresh arrays): what ``LeoPowerAttVecEnv.step_wait`` reads, which lays the
        observations out afresh anyway."""
        if not copy:
            pin = getattr(self, "_pin", None)
            if pin is None:
                from ... import _hip
                n = self.n_envs
                f64, u8 = _hip.PinnedBuffer(6 * n * 8), _hip.PinnedBuffer(max(n, 1))
                blk = f64.array.view(np.float64).reshape(6, n)
                pin = self._pin = {"bufs": [f64, u8], "obs": blk[:5], "rew": blk[5], "why": u8.array[:n],
                                   "p_obs": f64.ptr, "p_rew": f64.ptr + 5 * n * 8, "p_why": u8.ptr}
            check(self._lib.bsk_get_obs(self._handle(), pin["p_obs"], pin["p_rew"], None, pin["p_why"]))
            return pin["obs"], pin["rew"], pin["why"] != 0, pin["why"]
        obs = np.empty((5, self.n_envs), dtype=np.float64)
        rew = np.empty(self.n_envs, dtype=np.float64)
        done = np.empty(self.n_envs, dtype=np.uint8)
        why = np.empty(self.n_envs, dtype=np.uint8)
        check(self._lib.bsk_get_obs(self._handle(), obs.ctypes.data, rew.ctypes.data, done.ctypes.data, why.ctypes.data))
        return obs, rew, done.astype(bool), why

    def get_obs_rowmajor(self):
        """-> obs (N, 5) f64, reward (N,), done (N,) bool, reason (N,) uint8 as views of page-locked buffers the NEXT call overwrites:
        the row-major observation block the kernel writes under ``FLAG_OBS_ROWMAJOR``, in one contiguous copy (what a VecEnv hands
        out as (N, 5, 1) without a host-side transposition).  None without the flag."""
        if not (self.cfg.flags & FLAG_OBS_ROWMAJOR):
            return None
        pin = getattr(self, "_pin_rm", None)
        if pin is None:
            from ... import _hip
            n = self.n_envs
            f64, u8 = _hip.PinnedBuffer(6 * n * 8), _hip.PinnedBuffer(max(n, 1))
            flat = f64.array.view(np.float64)
            pin = self._pin_rm = {"bufs": [f64, u8], "obs": flat[:5 * n].reshape(n, 5), "rew": flat[5 * n:6 * n], "why": u8.array[:n],
                                  "p_obs": f64.ptr, "p_rew": f64.ptr + 5 * n * 8, "p_why": u8.ptr}
        check(self._lib.bsk_get_obs_rowmajor(self._handle(), pin["p_obs"], pin["p_rew"], pin["p_why"]))
        return pin["obs"], pin["rew"], pin["why"] != 0, pin["why"]

    def get_obs_state(self):
        """-> obs (5, N), state (n_fields, N) with one stream synchronisation (what the single-env mirror reads per step)."""
        obs = np.empty((5, self.n_envs), dtype=np.float64)
        st = np.empty((self.n_fields, self.n_envs), dtype=np.float64)
        check(self._lib.bsk_get_obs_state(self._handle(), obs.ctypes.data, None, None, st.ctypes.data))
        return obs, st

    def batch_stats(self):
        s, d = C.c_double(), C.c_int64()
        check(self._lib.bsk_get_batch_stats(self._handle(), C.byref(s), C.byref(d)))
        return s.value, d.value

    def stream_ptr(self):
        """The hipStream_t (as an integer) the handle launches on."""
        st = C.c_void_p()
        check(self._lib.bsk_get_stream(self._handle(), C.byref(st)))
        return st.value or 0

    def device_views(self):
        """Zero-copy device views (``__cuda_array_interface__`` + DLPack): obs (5, N) with the padded env stride,
        reward (N,), done_mask (ceil(N/64),) uint64, reason (N,) uint8, state (n_fields, N) strided; with a staged
        IC pool also terminal_obs (5, N) strided and episodes (N,) int32 of the device-side auto-reset."""
        po, pr, pm, pw, st = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self._lib.bsk_get_obs_device(self._handle(), C.byref(po), C.byref(pr), C.byref(pm), C.byref(pw), C.byref(st)))
        n = self.n_envs
        stream = self.stream_ptr()
        kw = {"owner": self, "device": self.device, "stream": stream}
        out = {
            "obs": _DevArray(po.value, (5, n), "<f8", strides=(st.value * 8, 8), **kw),
            "reward": _DevArray(pr.value, (n,), "<f8", **kw),
            "done_mask": _DevArray(pm.value, ((n + 63) // 64,), "<u8", **kw),
            "reason": _DevArray(pw.value, (n,), "|u1", **kw),
            "stride": st.value,
        }
        ps, sst = C.c_void_p(), C.c_int64()
        check(self._lib.bsk_get_state_device(self._handle(), C.byref(ps), C.byref(sst)))      # (the slab's rows have a stride of their own)
        out["state"] = _DevArray(ps.value, (self.n_fields, n), "<f8", strides=(sst.value * 8, 8), **kw)
        out["state_stride"] = sst.value
        pt, pe = C.c_void_p(), C.c_void_p()
        check(self._lib.bsk_get_terminal_obs_device(self._handle(), C.byref(pt), C.byref(pe)))
        if pt.value:
            out["terminal_obs"] = _DevArray(pt.value, (5, n), "<f8", strides=(st.value * 8, 8), **kw)
            out["episodes"] = _DevArray(pe.value, (n,), "<i4", **kw)
        # device-resident episode statistics / row-major observation (FLAG_EPISODE_STATS / FLAG_OBS_ROWMAJOR)
        er, tr, tl, dn, rm = (C.c_void_p() for _ in range(5))
        if hasattr(self._lib, "bsk_get_episode_device"):
            check(self._lib.bsk_get_episode_device(self._handle(), C.byref(er), C.byref(tr), C.byref(tl), C.byref(dn), C.byref(rm)))
        if er.value:
            out["episode_return"] = _DevArray(er.value, (n,), "<f8", **kw)
            out["terminal_return"] = _DevArray(tr.value, (n,), "<f8", **kw)
            out["terminal_length"] = _DevArray(tl.value, (n,), "<i4", **kw)
            out["done"] = _DevArray(dn.value, (n,), "|b1", **kw)
        if rm.value:
            out["obs_rowmajor"] = _DevArray(rm.value, (n, 5), "<f8", **kw)
        return out

    def get_ic_pool(self):
        """Host copy [n_fields, n_pool] of the staged pool (as set or as sampled on the device)."""
        if not getattr(self, "_n_pool", 0):
            raise RuntimeError("no IC pool staged")
        pool = np.empty((self.n_fields, self._n_pool), dtype=np.float64)
        check(self._lib.bsk_get_ic_pool(self._handle(), pool.ctypes.data))
        return pool

    def set_ic_pool(self, ic_pool):
        """Stage initial conditions [n_fields, n_pool] for device-side auto-reset (FLAG_AUTO_RESET)."""
        pool = np.ascontiguousarray(ic_pool, dtype=np.float64)
        if pool.ndim != 2 or pool.shape[0] != self.n_fields:
            raise ValueError("ic_pool must have shape (%d, n_pool)" % self.n_fields)
        check(self._lib.bsk_set_ic_pool(self._handle(), pool.shape[1], pool.ctypes.data))
        self._n_pool = pool.shape[1]

    def sample_ic_pool(self, n_pool, seed):
        """Draw ``n_pool`` initial conditions on the device (Philox4x32-10, reference distributions)."""
        check(self._lib.bsk_sample_ic_pool(self._handle(), int(n_pool), int(seed) & 0xFFFFFFFFFFFFFFFF))
        self._n_pool = int(n_pool)

    def reset_from_pool(self, mask=None):
        """(Re)start all (or the masked) envs from the staged pool, entirely on the device."""
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mp = mask.ctypes.data
        check(self._lib.bsk_reset_from_pool(self._handle(), mp))

    def reset_from_pool_device(self, d_mask_ptr=None):
        """The same with the mask (uint8[n], or None for every env) in DEVICE memory: enqueued on the handle's stream,
        no host data, no copy, no synchronisation."""
        check(self._lib.bsk_reset_from_pool_device(self._handle(), C.c_void_p(int(d_mask_ptr)) if d_mask_ptr else None))

    def batch_stats_device(self):
        """-> device pointer of f64[2] = {sum of rewards, number of done envs} of the last step, produced on the handle's
        stream without synchronising (the operand of a sharded batch's one all-reduce)."""
        p = C.c_void_p()
        check(self._lib.bsk_get_batch_stats_device(self._handle(), C.byref(p)))
        return p.value

    def set_step_stats(self, on=True):
        """Every step launch forms the per-64-env reward sums in its own epilogue, so that batch_stats() / batch_stats_device()
        behind it cost one small launch instead of two (same bits).  For loops that ask after every step; default off."""
        check(self._lib.bsk_set_step_stats(self._handle(), 1 if on else 0))

    def get_terminal_obs(self):
        """-> terminal observations (5, N) (valid where the last step reported done), finished-episode
        counts (N,) int32."""
        tob = np.empty((5, self.n_envs), dtype=np.float64)
        eps = np.empty(self.n_envs, dtype=np.int32)
        check(self._lib.bsk_get_terminal_obs(self._handle(), tob.ctypes.data, eps.ctypes.data))
        return tob, eps

    def sync(self):
        check(self._lib.bsk_sync(self._handle()))

    def set_sim_time(self, t):
        check(self._lib.bsk_set_sim_time(self._handle(), float(t)))

    def set_env_base(self, base):
        """Global index of this handle's env 0 (sharded batches; the device-side reset hashes the global index)."""
        check(self._lib.bsk_set_env_base(self._handle(), int(base)))
        self.env_base = int(base)

    # ------------------------------------------------------------------ measurement
    @staticmethod
    def debug_counters():
        """-> (host <-> device copies, stream synchronisations) the library has issued in this process so far."""
        a, b = C.c_int64(), C.c_int64()
        check(_lib.load().bsk_debug_counters(C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_words(self):
        """Probe builds (csrc/bsk_probes.hpp): the 64-bit word every wavefront of the last launch left -> uint64[ceil(n / 64)]."""
        out = np.zeros((self.n_envs + 63) // 64, dtype=np.uint64)
        check(self._lib.bsk_debug_words(self._handle(), out.ctypes.data))
        return out

    def profile_begin(self, capacity, stride=1):
        """Arm dispatch-timestamp profiling of the step kernel for every ``stride``-th launch."""
        check(self._lib.bsk_profile_set_stride(self._handle(), int(stride)))
        check(self._lib.bsk_profile_begin(self._handle(), int(capacity)))

    def profile_end(self):
        ms, n = C.c_double(), C.c_int()
        check(self._lib.bsk_profile_end(self._handle(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_end_samples(self, cap=4096):
        """-> (mean kernel ms, individual kernel durations [ms] as a float32 array)."""
        ms, n = C.c_double(), C.c_int()
        buf = np.zeros(cap, dtype=np.float32)
        check(self._lib.bsk_profile_end_samples(self._handle(), C.byref(ms), C.byref(n), buf.ctypes.data, cap))
        return ms.value, buf[:min(cap, n.value)].copy()

    def kernel_info(self):
        name = C.create_string_buffer(128)
        v, l, b, g = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._lib.bsk_kernel_info(self._handle(), name, 128, C.byref(v), C.byref(l), C.byref(b), C.byref(g)))
        return {"name": name.value.decode(), "vgprs": v.value, "lds_bytes": l.value, "block": b.value, "grid": g.value}
