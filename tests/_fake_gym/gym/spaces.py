"""TEST STAND-IN: ``gym.spaces.Box`` / ``Discrete`` with the constructor signatures the reference uses
(envs/leoPowerAttitudeEnvironment.py:43-53)."""
import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super(Box, self).__init__(shape if shape is not None else np.shape(low), dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n):
        super(Discrete, self).__init__((), np.int64)
        self.n = int(n)

    def contains(self, x):
        return int(x) == x and 0 <= int(x) < self.n
