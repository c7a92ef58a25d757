"""basilisk_env_amd — MI355X-native batched spacecraft-dynamics gym environment.

Registers ``leo_power_att_env-v0`` like the reference package does
(reference basilisk_env/__init__.py:6-9) when gym is importable.
"""
import logging

logger = logging.getLogger(__name__)

__version__ = "0.1.0"

ENV_ID = 'leo_power_att_env-v0'
ENTRY_POINT = 'basilisk_env_amd.envs:leoPowerAttEnv'

# (gym is absent from the build image: tests/test_gym_boundary.py executes this branch against a stand-in package)
try:
    from gym.envs.registration import register
except ImportError:
    register = None
if register is not None:
    try:
        register(
            id=ENV_ID,
            entry_point=ENTRY_POINT
        )
    except Exception as e:  # the id is taken (the reference package was imported first: it registers the same id)
        logger.warning("gym id %s not registered for %s: %r", ENV_ID, ENTRY_POINT, e)
