"""TEST STAND-IN: the abstract ``VecEnv`` with the method list of stable-baselines 2.10 (reset, step_async, step_wait, close,
get_attr, set_attr, env_method, seed abstract; step / render / unwrapped / _get_indices concrete) plus stable-baselines3's
``env_is_wrapped`` - the union, so that a class instantiable against this base is instantiable against either."""
from abc import ABC, abstractmethod


class VecEnv(ABC):
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    @abstractmethod
    def close(self):
        pass

    @abstractmethod
    def get_attr(self, attr_name, indices=None):
        pass

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None):
        pass

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        pass

    @abstractmethod
    def seed(self, seed=None):
        pass

    @abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None):
        pass

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self

    def _get_indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)


class VecEnvWrapper(VecEnv):
    """The wrapper base stable-baselines' VecNormalize / VecMonitor derive from: delegates to ``venv``."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        VecEnv.__init__(self, num_envs=venv.num_envs, observation_space=observation_space or venv.observation_space,
                        action_space=action_space or venv.action_space)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def reset(self):
        return self.venv.reset()

    def step_wait(self):
        return self.venv.step_wait()

    def close(self):
        return self.venv.close()

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self.venv.set_attr(attr_name, value, indices)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return self.venv.env_method(method_name, *method_args, indices=indices, **method_kwargs)

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return self.venv.env_is_wrapped(wrapper_class, indices=indices)
