"""leoPowerAttEnv — gym.Env surface of the reference's LEO power/attitude environment, stepping
the HIP propagator instead of the Basilisk engine.

Mirrors reference ``envs/leoPowerAttitudeEnvironment.py`` member for member: constructor
constants (:20-59), ``step`` (:65-145), ``_take_action`` (:147-159), ``_get_reward`` (:161-170),
``reset`` (:172-191), ``reset_init`` (:202-216).  Old gym API: ``step`` returns a 4-tuple,
observations are ``(5,1)`` float64.
"""
import copy
import logging

import numpy as np

from .. import spaces
from ..simulators import leoPowerAttitudeSimulator
from ..simulators.dynamics.config import RPM

logger = logging.getLogger(__name__)


class leoPowerAttEnv(spaces.Env):
    """Simple attitude/orbit control problem: point at the ground (reward) or at the Sun (power)."""

    def __init__(self, simulator_kwargs=None):
        self.__version__ = "0.1.0"
        logger.info("Basilisk Attitude Mode Management Sim (HIP propagator) - Version %s", self.__version__)

        self.max_length = int(3 * 180)

        self.simulator_init = 0
        self.simulator = None
        self.simulator_backup = None
        self.reward_total = 0

        self.mass = 330.0  # kg
        self.powerDraw = -5.  # W
        self.wheel_limit = 3000 * RPM  # 3000 RPM in radians/s
        self.power_max = 20.0  # W/Hr

        self.step_duration = 180.
        self.reward_mult = 1. / self.max_length
        self.failure_penalty = 1
        low = -1e16
        high = 1e16
        self.observation_space = spaces.Box(low, high, shape=(5, 1))
        self.obs = np.zeros([5, ])
        self.debug_states = []
        self.sim_over = False

        #   0 - earth pointing, 1 - sun pointing, 2 - desaturation
        self.action_space = spaces.Discrete(3)

        self.curr_episode = -1
        self.action_episode_memory = []
        self.curr_step = 0
        self.episode_over = False
        # extra (not in the reference): forwarded to the simulator (n_rw, gravity_model, device, ...)
        self._simulator_kwargs = dict(simulator_kwargs or {})

    def _make_simulator(self, initial_conditions=None):
        return leoPowerAttitudeSimulator.LEOPowerAttitudeSimulator(.1, 1.0, self.step_duration, initial_conditions,
                                                                   **self._simulator_kwargs)

    def seed(self, seed=None):
        """Seeds the legacy numpy RNG the IC samplers draw from (the reference inherits gym's
        no-op ``seed``, so its ICs are never reproducible; :61-63 is dead code there)."""
        if seed is not None:
            np.random.seed(seed)
        return [seed]

    def step(self, action):
        if self.simulator_init == 0:
            # the reference builds the simulator lazily here with kwargs its class rejects (:95),
            # so reset() is effectively mandatory; keep that contract but say so
            raise RuntimeError("call reset() before step()")

        if self.curr_step >= self.max_length:
            self.episode_over = True

        prev_ob = self._get_state()
        self._take_action(action)

        reward = self._get_reward()
        self.reward_total += reward
        ob = self._get_state()
        ob[2] = ob[2] / self.wheel_limit
        ob[3] = ob[3] / self.power_max
        if ob[2] > 1:
            self.episode_over = True
            reward -= self.failure_penalty
            self.reward_total -= self.failure_penalty
            logger.info("Died from wheel explosion. RPMs were norm: %s, limit is %s, body rate was %s, action taken was %s, env step %s",
                        ob[2] * self.wheel_limit, self.wheel_limit, ob[1], action, self.curr_step)
            logger.info("Prior state was RPM: %s . body rate was: %s", prev_ob[2] * self.wheel_limit, prev_ob[1])

        if ob[3] == 0:
            self.episode_over = True
            reward -= self.failure_penalty
            self.reward_total -= self.failure_penalty
            logger.info("Ran out of power. Battery level was at: %s, env step %s", prev_ob[3], self.curr_step - 1)

        if self.sim_over:
            self.episode_over = True
            logger.info("Orbit decayed - no penalty, but this one is over.")

        if self.episode_over:
            info = {'episode': {'r': self.reward_total, 'l': self.curr_step},
                    'full_states': self.debug_states,
                    'obs': ob}
            self.simulator.close_gracefully()
        else:
            info = {'full_states': self.debug_states, 'obs': ob}

        self.curr_step += 1
        return ob, reward, self.episode_over, info

    def _take_action(self, action):
        self.action_episode_memory[self.curr_episode].append(action)
        self.obs, self.debug_states, self.sim_over = self.simulator.run_sim(action)

    def _get_reward(self):
        """Nadir-pointing quality, only when the nadir mode was commanded (:161-170)."""
        reward = 0
        if self.action_episode_memory[self.curr_episode][-1] == 0:
            reward = np.linalg.norm(self.reward_mult / (1. + self.obs[0] ** 2.0))
        return reward

    def reset(self):
        self.action_episode_memory.append([])
        self.episode_over = False
        self.curr_step = 0
        self.reward_total = 0
        self._drop_simulator()
        self.simulator = self._make_simulator()
        self.simulator_init = 1
        ob = copy.deepcopy(self.simulator.obs)
        ob[2] = ob[2] / self.wheel_limit
        ob[3] = ob[3] / self.power_max
        return ob

    def _drop_simulator(self):
        sim, self.simulator = self.simulator, None
        if sim is not None and hasattr(sim, "release"):
            sim.release()      # the device handle is parked and re-used by the next simulator

    def _render(self, mode='human', close=False):
        return

    def _get_state(self):
        return self.simulator.obs

    def reset_init(self):
        """Restart the episode from the current simulator's initial conditions (:202-216)."""
        self.action_episode_memory.append([])
        self.episode_over = False
        self.curr_step = 0
        self.reward_total = 0
        initial_conditions = self.simulator.initial_conditions
        self._drop_simulator()
        self.simulator = self._make_simulator(initial_conditions)
        self.simulator_init = 1
        ob = copy.deepcopy(self.simulator.obs)
        ob[2] = ob[2] / self.wheel_limit
        ob[3] = ob[3] / self.power_max
        return ob

    def close(self):
        self._drop_simulator()
        leoPowerAttitudeSimulator.drain_idle_propagators()



def make_env(**kwargs):
    """The environment the way the reference's main obtains it (:219): ``gym.make('leo_power_att_env-v0')`` through the id this
    package registers when gym imports; the class itself where gym is absent."""
    if spaces.HAVE_GYM:
        import gym

        from .. import ENV_ID
        return gym.make(ENV_ID, **kwargs)
    return leoPowerAttEnv(**kwargs)


def demo(episodes=2, action=0, seed=12345, plot=False, env_kwargs=None):
    """What the reference module does when run as a script (:218-244): roll whole episodes with one fixed
    action and keep the observation history.  Returns the per-episode histories (5, steps); prints one
    line per episode; ``plot=True`` draws them if matplotlib is available."""
    names = ("attitude error", "body rate", "wheel speed / limit", "battery / capacity", "sunlit fraction")
    env = make_env(**(env_kwargs or {}))
    histories = []
    for ep in range(episodes):
        env.reset()
        env.seed(seed=seed)
        rows, ret = [], 0.0
        while True:
            ob, reward, over, _ = env.step(action)
            rows.append(ob[:, 0].copy())
            ret += reward
            if over or len(rows) >= env.max_length:
                break
        hist = np.array(rows).T
        histories.append(hist)
        print("episode %d: %d steps, return %.4f, last observation %s" % (ep, hist.shape[1], ret, np.array2string(hist[:, -1], precision=5)))
    env.close()
    if plot:
        from matplotlib import pyplot as plt
        for ep, hist in enumerate(histories):
            plt.figure()
            for k, name in enumerate(names):
                plt.plot(hist[k], label=name)
            plt.grid()
            plt.legend()
            plt.title("episode %d" % ep)
        plt.show()
    return histories


def demo_batch(num_envs=1024, episodes=2, action=0, seed=12345, **vec_env_kwargs):
    """The script's loop (:218-231: whole episodes of one action) for a BATCH of spacecraft: one ``LeoPowerAttVecEnv.rollout`` call per
    episode - ``max_length + 1`` env steps enqueued by one library call, no host visit per step - instead of 541 ``step`` calls.
    -> per episode (observations (steps, num_envs, 5), returns (num_envs,), lengths (num_envs,)) up to each spacecraft's first done;
    prints one line per episode."""
    from .leoPowerAttitudeVecEnv import LeoPowerAttVecEnv
    env = LeoPowerAttVecEnv(num_envs, seed=seed, auto_reset=False, **vec_env_kwargs)
    out = []
    for ep in range(episodes):
        env.reset()
        obs, rew, dones, _ = env.rollout(env.max_length + 1, constant_action=action)
        first = np.where(dones.any(axis=0), dones.argmax(axis=0), dones.shape[0] - 1)          # each spacecraft's terminal step
        alive = np.arange(dones.shape[0])[:, None] <= first[None, :]
        ret = (rew * alive).sum(axis=0)
        out.append((obs[..., 0], ret, first + 1))
        print("episode %d: %d spacecraft, steps %d ... %d, mean return %.4f" % (ep, num_envs, (first + 1).min(), (first + 1).max(), ret.mean()))
    env.close()
    return out


if __name__ == "__main__":
    import sys
    demo(plot="--plot" in sys.argv)
