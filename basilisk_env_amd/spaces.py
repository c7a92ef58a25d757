"""Tiny stand-ins for ``gym.spaces.Box`` / ``Discrete`` used when gym is not installed.

The reference builds ``spaces.Box(-1e16, 1e16, shape=(5,1))`` and ``spaces.Discrete(3)``
(envs/leoPowerAttitudeEnvironment.py:43-53); when gym imports, the real classes are used.
"""
import numpy as np

try:  # (executed against a stand-in gym by tests/test_gym_boundary.py)
    from gym import Env as _GymEnv
    from gym.spaces import Box, Discrete
    HAVE_GYM = True
except Exception:  # ModuleNotFoundError or a broken install
    HAVE_GYM = False

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def __repr__(self):
            return "Box%r" % (self.shape,)

    class Discrete(object):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x):
            return int(x) == x and 0 <= int(x) < self.n

        def sample(self):
            return int(np.random.randint(self.n))

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class _GymEnv(object):
        metadata = {"render.modes": []}
        reward_range = (-float("inf"), float("inf"))
        action_space = None
        observation_space = None

        def seed(self, seed=None):
            return [seed]

        def render(self, mode="human"):
            return None

        def close(self):
            return None

Env = _GymEnv
