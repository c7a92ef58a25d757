"""TEST STAND-IN for the ``gym`` package (absent from the build image): just enough of the old-API surface for the reference's way in -
``gym.make('leo_power_att_env-v0')`` after ``register(id=..., entry_point='module:attr')`` - to execute.  Put on sys.path by
tests/test_gym_boundary.py in a subprocess only; the product never sees it."""
from . import spaces  # noqa: F401
from .envs.registration import make, register, registry  # noqa: F401


class Env(object):
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return

    @property
    def unwrapped(self):
        return self
