from .leoPowerAttitudeEnvironment import leoPowerAttEnv  # noqa: F401
from .leoPowerAttitudeVecEnv import LeoPowerAttVecEnv  # noqa: F401
