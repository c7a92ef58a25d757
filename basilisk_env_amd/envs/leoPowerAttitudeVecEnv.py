"""LeoPowerAttVecEnv — N independent LEO power/attitude environments stepped by one HIP launch.

The batched sibling of ``leoPowerAttEnv``: the same observation (5 entries, normalised as in
reference envs/leoPowerAttitudeEnvironment.py:107-108), reward (:161-170) and termination
(:98-127) per env, with the stable-baselines ``VecEnv`` calling convention
(``reset() -> (N,5,1)``, ``step_async``/``step_wait``, ``step(actions) -> (obs, rews, dones,
infos)``, auto-reset of finished envs, ``get_attr``/``set_attr``/``env_method``).  Reward and
done flags are computed on the device by the step kernel.
"""
import numpy as np

from .. import spaces
from .._lib import (FLAG_AUTO_RESET, FLAG_DESAT, FLAG_DRAG, FLAG_EPISODE_STATS, FLAG_OBS_ROWMAJOR, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2,
                     DONE_BATTERY, DONE_LENGTH, DONE_ORBIT, DONE_WHEELS)
from ..simulators.dynamics.config import default_config
from ..simulators.dynamics.propagator import BatchedPropagator
from ..simulators.initial_conditions.batch import sample_ic_batch



def _vec_env_base():
    """stable-baselines' abstract ``VecEnv`` (so that ``isinstance`` checks in its wrappers hold) when the caller's
    process already uses it — the module is imported — or asks for it with ``BSKGPU_SB_VECENV=1``; else ``object``.
    Importing this package never pulls in stable-baselines (and with it torch / tensorflow) on its own; the class
    below implements the whole protocol itself either way."""
    import importlib
    import os
    import sys
    want = os.environ.get("BSKGPU_SB_VECENV") == "1"
    for mod in ("stable_baselines.common.vec_env", "stable_baselines3.common.vec_env"):
        if not want and mod not in sys.modules:
            continue
        try:
            return importlib.import_module(mod).VecEnv
        except Exception:
            continue
    return object


_Base = _vec_env_base()

# batches up to this size get a plain list of dicts as ``infos`` (every consumer sees exactly the old contract);
# above it every env's dict - empty or terminal - is made on first access (InfoList)
INFO_LAZY_ABOVE = 4096


class TerminalRecords(object):
    """What the finished envs of one step report, kept as ARRAYS: ``idx`` (ascending env indices), the finished episodes'
    returns ``r`` and lengths ``l`` (the Monitor convention of reference envs/leoPowerAttitudeEnvironment.py:130-135), their
    last observations ``obs`` (5, m) and done reasons ``why`` (m,).  ``info(i)`` builds env i's dict on demand: with the
    reference's ``max_length = 540`` and a common ``reset()`` EVERY env of a batch finishes on the same step, and building
    65 536 dicts (plus as many (5,1) arrays) eagerly stalled that step for 0.3 - 0.4 s."""

    __slots__ = ("idx", "r", "l", "obs", "why")

    def __init__(self, idx, r, l, obs, why):
        """``idx`` None: the arrays are DENSE - one entry per env of the batch, valid where ``why`` is non-zero (what a step
        where most of the batch finishes keeps: whole-array copies instead of 65 536-element gathers)."""
        self.idx, self.r, self.l, self.obs, self.why = idx, r, l, obs, why

    def position(self, i):
        """-> position of env ``i`` in the arrays if it finished at this step, else -1."""
        if self.idx is None:
            return i if self.why[i] else -1
        k = int(np.searchsorted(self.idx, i))
        return k if k < self.idx.size and self.idx[k] == i else -1

    def info(self, k):
        w = int(self.why[k])
        return {
            "episode": {"r": float(self.r[k]), "l": int(self.l[k])},
            "terminal_observation": self.obs[:, k].reshape(5, 1).copy(),
            "done_reason": {"length": bool(w & DONE_LENGTH), "wheels": bool(w & DONE_WHEELS),
                            "battery": bool(w & DONE_BATTERY), "orbit": bool(w & DONE_ORBIT)},
        }


class InfoList(list):
    """``infos`` of one step: a list of one dict per env whose dicts come into being when they are first looked at.

    Building 65 536 dicts per step takes longer (4 ms for empty ones, 0.3 s for terminal ones) than the step kernel.  Slots
    hold ``None`` internally and are replaced by a fresh dict - one per env, never shared - the first time the slot is read
    through indexing, slicing or iteration: ``{}`` for an env with nothing to report, the terminal info (``episode``,
    ``terminal_observation``, ``done_reason``) built from the step's ``TerminalRecords`` for a finished one.  Wrappers that
    read, copy or write into ``infos[i]`` see exactly the list-of-dicts they expect.
    (Consumers that read a list's storage from C without going through ``__getitem__`` / ``__iter__`` - ``json.dumps``,
    ``numpy.array``, ``pandas.DataFrame`` - would see ``None`` in untouched slots: hand them ``list(infos)``.)"""

    __slots__ = ("_term",)

    def __init__(self, n, terminal=None):
        list.__init__(self, [None]) if n == 1 else list.__init__(self, [None] * n)
        self._term = terminal

    def _fill(self, i):
        d = {}
        t = self._term
        if t is not None:
            k = t.position(i if i >= 0 else i + len(self))
            if k >= 0:
                d = t.info(k)
        list.__setitem__(self, i, d)
        return d

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        d = list.__getitem__(self, i)
        return self._fill(i) if d is None else d

    def __iter__(self):
        for i in range(len(self)):
            d = list.__getitem__(self, i)
            yield self._fill(i) if d is None else d

    def __reversed__(self):
        for i in range(len(self) - 1, -1, -1):
            yield self[i]

    def __contains__(self, item):
        return any(d == item for d in self)

    def __eq__(self, other):
        return list(self) == list(other)

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __repr__(self):
        return repr(list(self))

    def __reduce__(self):                      # pickle / copy / deepcopy: as the plain list of dicts it stands for
        return (list, (list(self),))

    def copy(self):
        return list(self)

    def __add__(self, other):
        return list(self) + list(other)

    def __mul__(self, k):
        raise TypeError("infos hold one dict per env; repeating the list would share them")

    __rmul__ = __imul__ = __mul__


def pool_slot(env, episode, n_pool):
    """IC-pool slot the device-side reset picks for ``env`` after ``episode`` finished episodes
    (include/bskgpu.h: bsk_set_ic_pool)."""
    return ((int(env) * 2654435761 + int(episode) * 40503 + 12345) & 0xFFFFFFFF) % int(n_pool)


def pool_slots(envs, episodes, n_pool):
    """``pool_slot`` for arrays of env indices and episode counts (the same rule in uint64 arithmetic)."""
    e = np.asarray(envs, dtype=np.uint64)
    k = np.asarray(episodes, dtype=np.int64).astype(np.uint64)          # (-1 wraps like the device's unsigned arithmetic)
    return (((e * np.uint64(2654435761) + k * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)) % np.uint64(int(n_pool))).astype(np.int64)


class LeoPowerAttVecEnv(_Base):
    def __init__(self, num_envs, n_rw=4, gravity_model=GRAV_PM_J2, step_duration=180., dynRate=0.1, fswRate=1.0,
                 seed=0, device=0, cfg=None, auto_reset=True, propagator_factory=None, power=True, sun_third_body=True, drag=True, desat=True,
                 device_reset_pool=0, device_sampler=False, stream=None):
        """``device_reset_pool`` > 0 stages that many random initial conditions on the device and lets
        the step kernel reset finished envs itself (no host round trip at episode boundaries); 0 keeps
        the host-side masked reset with freshly sampled ICs.  With ``device_sampler`` the pool itself is
        drawn on the GPU (Philox4x32-10 keyed by ``seed``) and ``reset()`` restarts from it on the device.
        ``stream``: a hipStream_t (integer) the propagator launches on instead of a stream of its own — e.g.
        ``torch.cuda.current_stream().cuda_stream`` — so that ``step_tensors`` needs no cross-stream ordering."""
        if _Base is not object:
            _Base.__init__(self, int(num_envs), spaces.Box(-1e16, 1e16, shape=(5, 1)), spaces.Discrete(3))
        self.num_envs = int(num_envs)
        self.observation_space = spaces.Box(-1e16, 1e16, shape=(5, 1))
        self.action_space = spaces.Discrete(3)
        self.auto_reset = bool(auto_reset)
        if cfg is None:
            cfg = default_config(n_rw=n_rw, gravity_model=gravity_model)
            cfg.dt = float(dynRate)
            cfg.fsw_every = int(round(fswRate / dynRate))
            if power:
                cfg.flags |= FLAG_POWER
                if sun_third_body:
                    cfg.flags |= FLAG_SUN_THIRD_BODY
                if drag:
                    cfg.flags |= FLAG_DRAG
                if desat and n_rw:
                    cfg.flags |= FLAG_DESAT
            if device_reset_pool:
                # the device-resident configuration: the kernel restarts finished envs, keeps the episode statistics and
                # writes the observation row-major as well (what step_tensors hands to an on-GPU policy)
                cfg.flags |= FLAG_AUTO_RESET | FLAG_EPISODE_STATS | FLAG_OBS_ROWMAJOR
        self.cfg = cfg
        self.n_rw = int(cfg.n_rw)
        self.max_length = int(cfg.max_length)
        self.step_duration = float(step_duration)
        self.substeps = int(round(step_duration / cfg.dt))
        self.wheel_limit = cfg.wheel_limit
        self.power_max = cfg.power_max
        self.reward_mult = cfg.reward_mult
        self.failure_penalty = cfg.failure_penalty
        self._rng = np.random.Generator(np.random.PCG64(seed))
        kw = {"stream": stream} if stream else {}
        self.propagator = (propagator_factory or BatchedPropagator)(cfg, self.num_envs, device=device, **kw)
        self._actions = None
        self.device_reset = bool(cfg.flags & FLAG_AUTO_RESET)
        self._ic = None
        if self.device_reset:
            n_pool = int(device_reset_pool) or self.num_envs
            self.device_sampler = bool(device_sampler)
            if self.device_sampler:
                self.propagator.sample_ic_pool(n_pool, seed)
                self._pool = self.propagator.get_ic_pool()      # host mirror: reset_init() replays device-side resets
            else:
                self._pool = sample_ic_batch(n_pool, self.n_rw, rng=self._rng)
                self.propagator.set_ic_pool(self._pool)
        else:
            self.device_sampler = False
            self._pool = None
        self.episode_returns = np.zeros(self.num_envs)
        self.episode_lengths = np.zeros(self.num_envs, dtype=np.int64)

    # ------------------------------------------------------------------ VecEnv API
    def seed(self, seed=None):
        self._rng = np.random.Generator(np.random.PCG64(seed))
        return [seed] * self.num_envs

    def _initial_obs(self, ic, cols=None):
        """Observation of freshly reset envs: [|sigma_BN|, |omega|, |Omega|/limit, charge/3600/max, 1]
        (the reference reports |sigma_BN| at reset too, :348; wheel speeds here are in rad/s,
        consistently with every later step)."""
        sel = slice(None) if cols is None else cols
        t = 12 + self.n_rw
        ob = np.empty((5, ic[:, sel].shape[1]))
        ob[0] = np.linalg.norm(ic[6:9, sel], axis=0)
        ob[1] = np.linalg.norm(ic[9:12, sel], axis=0)
        ob[2] = np.linalg.norm(ic[12:12 + self.n_rw, sel], axis=0) / self.wheel_limit if self.n_rw else 0.0
        ob[3] = ic[t + 7, sel] / 3600. / self.power_max
        ob[4] = 1.0
        return ob

    def reset(self, ic=None):
        """Reset every env (fresh random ICs unless ``ic`` [n_fields, N] is given) -> obs (N,5,1)."""
        if ic is None and self.device_sampler:
            self.propagator.reset_from_pool()           # no host data involved
            self._ic = self.propagator.get_state()
        else:
            self._ic = sample_ic_batch(self.num_envs, self.n_rw, rng=self._rng) if ic is None else np.array(ic, dtype=np.float64)
            self.propagator.reset(self._ic)
        self._actions = None
        self.episode_returns[:] = 0
        self.episode_lengths[:] = 0
        return self._initial_obs(self._ic).T.reshape(self.num_envs, 5, 1)

    # The running episodes' initial conditions (what reset_init() replays).  Envs the DEVICE restarted are only marked: their
    # columns are filled in from the pool - by the slot rule of include/bskgpu.h: bsk_set_ic_pool - when somebody reads ``_ic``,
    # not at the step where they finished (gathering 65 536 pool columns costs more than ten kernel launches of that batch).
    @property
    def _ic(self):
        base = self._ic_base
        if base is not None and self._ic_stale is not None:
            idx = np.flatnonzero(self._ic_stale)
            if idx.size:
                _, episodes = self.propagator.get_terminal_obs()
                off = int(getattr(self.propagator, "env_base", 0))
                base[:, idx] = np.take(self._pool, pool_slots(off + idx, episodes[idx].astype(np.int64) - 1, self._pool.shape[1]), axis=1)
                self._ic_stale[idx] = False
        return base

    @_ic.setter
    def _ic(self, value):
        self._ic_base = value
        if value is None or not self.device_reset:
            self._ic_stale = None
        else:
            self._ic_stale = np.zeros(self.num_envs, dtype=bool)

    def _ic_from_pool(self):
        """The running episodes' initial conditions after device-side resets nobody mirrored on the host (reset_tensors /
        step_tensors): the pool slots the device picked (slot rule of include/bskgpu.h: bsk_set_ic_pool)."""
        _, episodes = self.propagator.get_terminal_obs()
        base = int(getattr(self.propagator, "env_base", 0))
        cols = pool_slots(base + np.arange(self.num_envs), episodes.astype(np.int64) - 1, self._pool.shape[1])
        return np.ascontiguousarray(np.take(self._pool, cols, axis=1))

    def reset_init(self):
        """Replay the current initial conditions (reference reset_init, :202-216)."""
        ic = self._ic
        if ic is None:
            ic = self._ic_from_pool()
        return self.reset(ic)

    def step_async(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.num_envs)
        if a.min() < 0 or a.max() > 2:
            raise ValueError("actions must be in {0, 1, 2}")
        self._actions = a
        self.propagator.step(a, self.substeps)

    def step_wait(self):
        """-> obs (N,5,1), rewards (N,), dones (N,) bool, infos (one dict per env).  Finished envs report ``episode`` {'r', 'l'}
        (reference envs/leoPowerAttitudeEnvironment.py:130-135), ``terminal_observation`` and ``done_reason`` and are restarted
        (by the step kernel itself with ``device_reset_pool``, else from fresh host-sampled initial conditions).  Everything per
        finished env is ARRAY work here - the step where all 65 536 episodes of a batch end together (every 541st step with the
        reference's ``max_length`` and a common ``reset()``) costs what any other step costs; the terminal dicts are built when
        ``infos[i]`` is first read (``InfoList`` / ``TerminalRecords``)."""
        if self._actions is None:
            raise RuntimeError("step_wait() without a pending step_async()")
        self._actions = None
        rm = self.propagator.get_obs_rowmajor() if self.device_reset and hasattr(self.propagator, "get_obs_rowmajor") else None
        if rm is not None:
            # the kernel's own row-major (N, 5) block in one contiguous copy: no transposition on the host (views of page-locked
            # buffers the next step overwrites - everything this method returns is laid out afresh)
            obs_nm, rew, done, why = rm
            rew = rew.copy()
            obs = None
            obs_out = obs_nm.reshape(self.num_envs, 5, 1).copy()
        elif getattr(self.propagator, "pinned_read_back", False):
            obs, rew, done, why = self.propagator.get_obs(copy=False)
            rew = rew.copy()
            obs_out = obs.T.reshape(self.num_envs, 5, 1).copy()
        else:                                          # sharded engine (pinned fan-in of its own), the tests' oracle stand-in
            obs, rew, done, why = self.propagator.get_obs()
            obs_out = obs.T.reshape(self.num_envs, 5, 1).copy()
        self.episode_returns += rew
        idx = np.flatnonzero(done)
        terminal = None
        if idx.size:
            dense = 4 * idx.size > self.num_envs       # most of the batch: whole-array copies beat 65 536-element gathers
            if self.device_reset:
                # the kernel already reset these envs and wrote their new first observation into obs
                term, _ = self.propagator.get_terminal_obs()       # (fresh arrays)
                tobs = term if dense else term[:, idx]
                if self._ic_stale is not None:         # (None: device-resident steps in between; rebuilt on demand, _ic_from_pool)
                    self._ic_stale[idx] = True
            else:
                tobs = obs.copy() if dense else obs[:, idx]
            if dense:
                terminal = TerminalRecords(None, self.episode_returns.copy(), self.episode_lengths.copy(), tobs, why.copy())
            else:
                terminal = TerminalRecords(idx, self.episode_returns[idx], self.episode_lengths[idx], tobs, why[idx])
            if not self.device_reset and self.auto_reset:
                fresh = sample_ic_batch(idx.size, self.n_rw, rng=self._rng)
                if idx.size == self.num_envs:          # the whole batch: no mask, no compaction in the library either
                    self._ic = fresh
                    self.propagator.reset(fresh)
                else:
                    self._ic_base[:, idx] = fresh
                    mask = np.zeros(self.num_envs, dtype=np.uint8)
                    mask[idx] = 1
                    self.propagator.reset(self._ic_base, mask=mask)
                obs_out[idx] = self._initial_obs(fresh).T.reshape(idx.size, 5, 1)
            if self.device_reset or self.auto_reset:
                self.episode_returns[idx] = 0
                self.episode_lengths[idx] = -1
        # one dict per env, made when first read; batches up to INFO_LAZY_ABOVE hand out the plain list of dicts
        infos = InfoList(self.num_envs, terminal)
        if self.num_envs <= INFO_LAZY_ABOVE:
            infos = list(infos)
        self.episode_lengths += 1
        return obs_out, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def rollout(self, n_steps, actions=None, constant_action=0):
        """Open-loop evaluation: ``n_steps`` env steps under a fixed action sequence (``actions`` int (n_steps, N)) or ONE action -
        what the reference's own mains do with one env (envs/leoPowerAttitudeEnvironment.py:218-231: whole episodes of action 0) -
        enqueued by one library call (``bsk_step_n``: one launch at the bare level, one launch per env step at the scenario levels)
        with no host visit per step.  -> obs (n_steps, N, 5, 1), rewards (n_steps, N), dones (n_steps, N) bool, reasons (n_steps, N)
        uint8: row t is what ``step()`` number t would have returned (with a device reset pool, the restarted envs' observation is
        the new episode's first one; without one, finished envs simply keep stepping - a host-side reset cannot happen inside the
        call, so ``auto_reset`` without ``device_reset_pool`` is refused).  The env's own bookkeeping (episode returns and lengths,
        the running episodes' initial conditions) ends as after those ``n_steps`` calls of ``step()``."""
        T = int(n_steps)
        if self.auto_reset and not self.device_reset:
            raise ValueError("rollout(): host-side auto-reset cannot run inside the call; use device_reset_pool or auto_reset=False")
        a = None
        if actions is not None:
            a = np.ascontiguousarray(actions, dtype=np.int32).reshape(T, self.num_envs)
            if a.min() < 0 or a.max() > 2:
                raise ValueError("actions must be in {0, 1, 2}")
        elif not 0 <= int(constant_action) <= 2:
            raise ValueError("actions must be in {0, 1, 2}")
        obs, rew, why = self.propagator.rollout(T, self.substeps, actions=a, constant_action=int(constant_action))
        dones = why != 0
        for t in range(T):                             # (array work per step: the bookkeeping of step_wait)
            self.episode_returns += rew[t]
            if self.device_reset:
                d = dones[t]
                if d.any():
                    self.episode_returns[d] = 0
                    self.episode_lengths[d] = -1
                    if self._ic_stale is not None:
                        self._ic_stale[d] = True
            self.episode_lengths += 1
        return obs.transpose(0, 2, 1)[..., None], rew, dones, why

    # ------------------------------------------------------------------ device-resident surface (row f4)
    def _torch_views(self):
        """torch tensors over the library's device buffers (zero copy, made once)."""
        tv = getattr(self, "_tviews", None)
        if tv is None:
            import torch
            dev = torch.device("cuda", self.propagator.device)
            v = self.propagator.device_views()
            tv = {"device": dev, "stream": self.propagator.stream_ptr()}
            for k in ("obs", "reward", "reason", "done_mask", "state", "terminal_obs", "episodes", "episode_return",
                      "terminal_return", "terminal_length", "obs_rowmajor", "done"):
                if k in v:
                    tv[k] = torch.as_tensor(v[k], device=dev)                # ("done": a bool view of the kernel's 0 / 1 byte)
            # (N,5,1): the row-major buffer when the kernel writes one (contiguous: a policy's reshape is free), else a
            # transposed view of the SoA buffer
            tv["obs_n51"] = tv["obs_rowmajor"].unsqueeze(-1) if "obs_rowmajor" in tv else tv["obs"].t().unsqueeze(-1)
            if "terminal_obs" in tv:
                tv["terminal_obs_n51"] = tv["terminal_obs"].t().unsqueeze(-1)
            tv["ext"] = None
            self._tviews = tv
        return tv

    def _order_streams(self, tv, before):
        """Device-side ordering between the caller's current torch stream and the handle's stream (nothing when they are the
        same stream): ``before`` = the handle's work waits for the caller's, else the caller's waits for the handle's."""
        import torch
        cur = torch.cuda.current_stream(tv["device"])
        if cur.cuda_stream == tv["stream"]:
            return
        if tv["ext"] is None:
            tv["ext"] = torch.cuda.ExternalStream(tv["stream"], device=tv["device"])
        if before:
            tv["ext"].wait_stream(cur)
        else:
            cur.wait_stream(tv["ext"])

    def reset_tensors(self):
        """``reset()`` for the device-resident loop -> the (N,5,1) device tensor of first observations, a zero-copy view of
        the buffer the reset kernel itself wrote.  With the pool drawn on the GPU (``device_sampler``) nothing crosses PCIe
        and nothing synchronises: one kernel on the handle's stream restarts every env from the pool.  Otherwise fresh
        host-sampled initial conditions are uploaded (the one unavoidable copy) and the observation is again the kernel's.
        Replaces reference envs/leoPowerAttitudeEnvironment.py:172-191 for an on-GPU training loop."""
        tv = self._torch_views()
        if self.device_sampler and hasattr(self.propagator, "reset_from_pool_device"):
            self._order_streams(tv, before=True)
            self.propagator.reset_from_pool_device(None)
            self._order_streams(tv, before=False)
            self._ic = None                     # reset_init() rebuilds it from the pool's slot rule on demand
            self._actions = None
            self.episode_returns[:] = 0
            self.episode_lengths[:] = 0
        else:
            self.reset()
        return tv["obs_n51"]

    def step_tensors(self, actions):
        """One env step with everything resident on the GPU: ``actions`` is an int32 or int64 device tensor (N,) - int64 is
        what ``argmax`` returns, read in place - and the result ``(obs (N,5,1), reward (N,), done (N,) bool, info)`` are
        device tensors: zero-copy views of the buffers the step kernel writes (valid until the next step).  This method
        launches NO torch kernel of its own, does not synchronise and moves nothing over PCIe: the launch is ordered after
        the producer of ``actions`` and before the consumers of the outputs on the device (same stream when the env was
        created on the caller's stream, stream waits otherwise).  Needs the device-side auto-reset (``device_reset_pool``)
        unless ``auto_reset=False``: finished envs are restarted by the kernel itself.  ``info`` holds device tensors:
        ``reason`` (N,) uint8; ``terminal_observation`` (N,5,1), ``episode_r`` (N,) and ``episode_l`` (N,) int32 - rows valid
        where ``done`` (the Monitor convention of reference envs/leoPowerAttitudeEnvironment.py:130-135, on the device);
        ``episode_return`` (N,), the running episodes' returns; ``episodes`` (N,) int32, finished episodes per env.
        The host-side mirrors (``episode_returns`` / ``episode_lengths``) are not updated by this path.
        Fastest with policy and env on ONE non-default torch stream (``stream=s.cuda_stream``, loop under
        ``torch.cuda.stream(s)``): torch's legacy default stream synchronises with every other stream of the process.
        Replaces reference envs/leoPowerAttitudeEnvironment.py:65-145 and simulators/leoPowerAttitudeSimulator.py:598-619
        for an on-GPU policy."""
        import torch
        if self.auto_reset and not self.device_reset:
            raise ValueError("step_tensors needs device_reset_pool > 0 (device-side auto-reset) or auto_reset=False")
        tv = self._torch_views()
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype in (torch.int32, torch.int64)
                and actions.is_contiguous() and actions.numel() == self.num_envs and actions.device == tv["device"]):
            raise ValueError("actions must be a contiguous int32 or int64 tensor of %d entries on %s" % (self.num_envs, tv["device"]))
        self._order_streams(tv, before=True)       # the kernel reads `actions` after their producer
        # `actions` stays referenced until the next step replaces it: by then the caller's stream has been ordered after
        # the kernel that read it (wait below), so the caching allocator may hand the block out again.  (No
        # record_stream on the handle's stream: the allocator would record an event on it when the tensor dies, possibly
        # after close() has destroyed that stream.)
        self._dev_actions = actions
        self.propagator.step_device(actions.data_ptr(), self.substeps, int64=actions.dtype == torch.int64)
        self._order_streams(tv, before=False)      # consumers on the caller's stream run after the kernel
        self._ic = None if self.device_reset else self._ic
        info = {"reason": tv["reason"]}
        if "terminal_obs_n51" in tv:
            info["terminal_observation"] = tv["terminal_obs_n51"]
            info["episodes"] = tv["episodes"]
        if "terminal_return" in tv:
            info["episode_r"], info["episode_l"], info["episode_return"] = tv["terminal_return"], tv["terminal_length"], tv["episode_return"]
        done = tv["done"] if "done" in tv else tv["reason"].ne(0)
        return tv["obs_n51"], tv["reward"], done, info

    def close(self):
        self._tviews = None            # torch views alias device buffers the propagator is about to free
        self._dev_actions = None
        self.propagator.close()

    def _n_indexed(self, indices):
        if indices is None:
            return self.num_envs
        idx = np.atleast_1d(np.asarray(indices, dtype=np.int64))
        if idx.size and (idx.min() < -self.num_envs or idx.max() >= self.num_envs):
            raise IndexError("env index out of range")
        return int(idx.size)

    def get_attr(self, attr_name, indices=None):
        """One value per selected env.  The envs share their constants (one batch, one config), so every entry
        is the batch's own attribute."""
        return [getattr(self, attr_name) for _ in range(self._n_indexed(indices))]

    def set_attr(self, attr_name, value, indices=None):
        """Attributes belong to the batch: setting one for a strict subset of the envs cannot be honoured."""
        if self._n_indexed(indices) != self.num_envs:
            raise ValueError("the envs of a LeoPowerAttVecEnv share their attributes; set_attr needs indices=None")
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """Calls the batch's method once and returns the result once per selected env."""
        n = self._n_indexed(indices)
        res = getattr(self, method_name)(*args, **kwargs)
        return [res for _ in range(n)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self._n_indexed(indices)

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode="human"):
        return None

    # ------------------------------------------------------------------ extras
    def get_state(self):
        return self.propagator.get_state()

    def batch_stats(self):
        """(sum of rewards, number of done envs) of the last step, reduced on the device."""
        return self.propagator.batch_stats()
