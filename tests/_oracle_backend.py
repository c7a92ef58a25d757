"""Oracle-backed stand-in for ``BatchedPropagator`` with the same Python interface.

TEST INFRASTRUCTURE: lets the CPU suite exercise the host logic (simulator mirror, gym env, vec
env, sharding) without a GPU, and gives the GPU suite a like-for-like checker.  It lives under
tests/ and is injected through ``propagator_factory``; the product never imports it.
"""
import numpy as np

from basilisk_env_amd._lib import n_fields
from oracle import oracle


class OraclePropagator(object):
    def __init__(self, cfg, n_envs, device=0, stream=None):
        self.cfg = cfg.copy()
        self.n_envs = int(n_envs)
        self.n_rw = int(cfg.n_rw)
        self.n_fields = n_fields(self.n_rw)
        self.state = np.zeros((self.n_fields, self.n_envs))
        self.steps = np.zeros(self.n_envs, np.int32)
        self.ticks = np.zeros(self.n_envs, np.int32)
        self._out = None
        self._cbar = self._sbar = None
        self._pool = None
        self._t0 = 0.0
        self.episodes = np.zeros(self.n_envs, np.int32)
        self._term_obs = np.zeros((5, self.n_envs))
        self.env_base = 0
        self.device = int(device)

    def set_gravity_sh(self, degree, cbar, sbar):
        assert degree == self.cfg.sh_degree
        self._cbar, self._sbar = np.array(cbar, dtype=np.float64), np.array(sbar, dtype=np.float64)

    def close(self):
        pass

    def reset(self, ic, mask=None):
        ic = np.asarray(ic, dtype=np.float64)
        assert ic.shape == self.state.shape
        if mask is None:
            self.state[:] = ic
            self.steps[:] = 0
            self.ticks[:] = 0
        else:
            m = np.asarray(mask).astype(bool)
            self.state[:, m] = ic[:, m]
            self.steps[m] = 0
            self.ticks[m] = 0

    def get_state(self):
        return self.state.copy()

    def set_state(self, state):
        self.state[:] = state

    def get_counters(self):
        return self.steps.copy(), self.ticks.copy()

    def set_counters(self, steps, ticks):
        self.steps[:] = steps
        self.ticks[:] = ticks

    def step(self, actions, substeps):
        self._out = oracle.step(self.cfg, self.state, self.steps, self.ticks, np.asarray(actions, np.int32), substeps,
                                sim_time0=self._t0, cbar=self._cbar, sbar=self._sbar)
        if self._pool is not None:
            self._auto_reset()

    def rollout(self, n_steps, substeps, actions=None, constant_action=0):
        """BatchedPropagator.rollout's contract, by its definition: the histories of n_steps single steps."""
        T, n = int(n_steps), self.n_envs
        obs, rew, why = np.empty((T, 5, n)), np.empty((T, n)), np.empty((T, n), dtype=np.uint8)
        for t in range(T):
            self.step(np.full(n, constant_action, np.int32) if actions is None else actions[t], substeps)
            obs[t], rew[t], _, why[t] = self.get_obs()
        return obs, rew, why

    def set_ic_pool(self, ic_pool):
        self._pool = np.array(ic_pool, dtype=np.float64)

    def sample_ic_pool(self, n_pool, seed):
        from _philox_ref import sample_pool
        self._pool = sample_pool(n_pool, self.n_rw, seed & 0xFFFFFFFFFFFFFFFF, mu=self.cfg.mu)

    def get_ic_pool(self):
        return self._pool.copy()

    def reset_from_pool(self, mask=None):
        n_pool = self._pool.shape[1]
        for i in range(self.n_envs):
            if mask is not None and not mask[i]:
                continue
            slot = (((i + self.env_base) * 2654435761 + int(self.episodes[i]) * 40503 + 12345) & 0xFFFFFFFF) % n_pool
            self.episodes[i] += 1
            self.state[:, i] = self._pool[:, slot]
            self.steps[i] = 0
            self.ticks[i] = 0

    def get_terminal_obs(self):
        return self._term_obs.copy(), self.episodes.copy()

    def _auto_reset(self):
        """Same rule as the step kernel's epilogue (include/bskgpu.h: bsk_set_ic_pool)."""
        obs, rew, done, why = self._out
        obs = obs.copy()
        n_pool = self._pool.shape[1]
        t = 12 + self.n_rw
        for i in np.flatnonzero(done):
            self._term_obs[:, i] = obs[:, i]
            slot = (((i + self.env_base) * 2654435761 + int(self.episodes[i]) * 40503 + 12345) & 0xFFFFFFFF) % n_pool
            self.episodes[i] += 1
            ic = self._pool[:, slot]
            self.state[:, i] = ic
            self.steps[i] = 0
            self.ticks[i] = 0
            wl = np.linalg.norm(ic[12:12 + self.n_rw]) / self.cfg.wheel_limit if self.n_rw else 0.0
            obs[:, i] = [np.linalg.norm(ic[6:9]), np.linalg.norm(ic[9:12]), wl, ic[t + 7] / 3600.0 / self.cfg.power_max, 1.0]
        self._out = (obs, rew, done, why)

    def get_obs(self):
        return self._out

    def batch_stats(self):
        return float(self._out[1].sum()), int(self._out[2].sum())

    def sync(self):
        pass

    def set_sim_time(self, t):
        self._t0 = float(t)

    def set_env_base(self, base):
        self.env_base = int(base)
