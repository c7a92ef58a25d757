"""basilisk_env_amd — MI355X-native batched spacecraft-dynamics gym environment.

Registers ``leo_power_att_env-v0`` like the reference package does
(reference basilisk_env/__init__.py:6-9) when gym is importable.
"""
import logging

logger = logging.getLogger(__name__)

__version__ = "0.1.0"

try:  # pragma: no cover - gym is absent in the build image
    from gym.envs.registration import register

    register(
        id='leo_power_att_env-v0',
        entry_point='basilisk_env_amd.envs:leoPowerAttEnv'
    )
except Exception:  # gym missing or the id already registered
    pass
