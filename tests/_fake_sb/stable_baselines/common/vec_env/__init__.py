from .base_vec_env import VecEnv, VecEnvWrapper  # noqa: F401
