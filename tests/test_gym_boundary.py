"""CPU: the gym-present / stable-baselines-present half of the drop-in boundary, EXECUTED ("What's missing" #4).

The reference's main reaches the env through the registered id - ``gym.make('leo_power_att_env-v0')``
(reference basilisk_env/envs/leoPowerAttitudeEnvironment.py:219; registration basilisk_env/__init__.py:6-9; agents README.md:14,22-27).
gym and stable-baselines are absent from the image, so ``register(...)`` (basilisk_env_amd/__init__.py), the ``from gym import Env`` /
``gym.spaces`` branch (spaces.py) and the ``VecEnv`` base-class selection (envs/leoPowerAttitudeVecEnv.py) never ran.  Here they run
against stand-in packages (tests/_fake_gym, tests/_fake_sb) in a SUBPROCESS, so that the package imports fresh with them on the path.
A mistyped entry-point string, an MRO clash with ``gym.Env`` or an unimplemented abstract method of ``VecEnv`` fails these tests.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(code, extra_env=None, fake_gym=True, fake_sb=False):
    paths = [ROOT, HERE]
    if fake_gym:
        paths.insert(0, os.path.join(HERE, "_fake_gym"))
    if fake_sb:
        paths.insert(0, os.path.join(HERE, "_fake_sb"))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(paths + [os.environ.get("PYTHONPATH", "")]))
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_register_make_and_step_through_the_gym_id():
    code = r"""
import json, numpy as np
import gym
from gym.envs import registration
import basilisk_env_amd
from basilisk_env_amd import spaces
from basilisk_env_amd.envs import leoPowerAttEnv
from _oracle_backend import OraclePropagator

out = {"calls": [[i, k] for i, k in registration.calls], "have_gym": spaces.HAVE_GYM}
env = gym.make('leo_power_att_env-v0', simulator_kwargs={"propagator_factory": OraclePropagator})
out["type_ok"] = type(env) is leoPowerAttEnv and isinstance(env, gym.Env) and spaces.Env is gym.Env
out["mro"] = [c.__name__ for c in type(env).__mro__]
out["spaces_ok"] = type(env.observation_space) is gym.spaces.Box and type(env.action_space) is gym.spaces.Discrete
out["obs_shape"] = list(env.observation_space.shape); out["n_actions"] = env.action_space.n
out["spec_id"] = env.spec.id
# the reference's main, shortened (envs/leoPowerAttitudeEnvironment.py:218-231): reset, seed, step action 0
ob0 = env.reset()
env.seed(seed=12345)
ob, reward, over, info = env.step(0)
out["reset_shape"] = list(ob0.shape); out["step_shape"] = list(ob.shape); out["dtype"] = str(ob.dtype)
out["reward"] = float(reward); out["over"] = bool(over); out["info_keys"] = sorted(info.keys())
out["obs_in_space"] = bool(env.observation_space.contains(ob))
env.close()
# a second registration of the id (the reference package imported as well) must not break the import
import importlib
importlib.reload(basilisk_env_amd)
out["calls_after_reload"] = len(registration.calls)
# demo() goes through gym.make when gym imports
from basilisk_env_amd.envs import leoPowerAttitudeEnvironment as E
made = []
orig = gym.make
def spy(id, **kw):
    made.append(id)
    return orig(id, **kw)
gym.make = spy
import basilisk_env_amd.envs.leoPowerAttitudeEnvironment as mod
e2 = mod.make_env(simulator_kwargs={"propagator_factory": OraclePropagator})
out["demo_via_make"] = made == ['leo_power_att_env-v0'] and type(e2) is leoPowerAttEnv
e2.close()
print(json.dumps(out))
"""
    out = _run(code)
    assert out["have_gym"] is True
    assert out["calls"] == [["leo_power_att_env-v0", {"entry_point": "basilisk_env_amd.envs:leoPowerAttEnv"}]]
    assert out["type_ok"] and out["spaces_ok"], out
    assert out["mro"] == ["leoPowerAttEnv", "Env", "object"]
    assert out["obs_shape"] == [5, 1] and out["n_actions"] == 3 and out["spec_id"] == "leo_power_att_env-v0"
    assert out["reset_shape"] == [5, 1] == out["step_shape"] and out["dtype"] == "float64"
    assert 0.0 < out["reward"] <= 1.0 / 540 and out["over"] is False and out["info_keys"] == ["full_states", "obs"] and out["obs_in_space"]
    assert out["calls_after_reload"] == 2          # attempted again, refused by the registry, import survived
    assert out["demo_via_make"]


def test_the_entry_point_string_is_what_make_resolves():
    """The registered string, resolved the way gym does (importlib + getattr), is the class - checked without any gym at all too,
    so that a typo cannot hide behind the import guard."""
    code = r"""
import importlib, json
import basilisk_env_amd
mod, attr = basilisk_env_amd.ENTRY_POINT.split(":")
cls = getattr(importlib.import_module(mod), attr)
from basilisk_env_amd import spaces
print(json.dumps({"name": cls.__name__, "module": cls.__module__, "id": basilisk_env_amd.ENV_ID, "have_gym": spaces.HAVE_GYM}))
"""
    out = _run(code, fake_gym=False)
    assert out == {"name": "leoPowerAttEnv", "module": "basilisk_env_amd.envs.leoPowerAttitudeEnvironment", "id": "leo_power_att_env-v0",
                   "have_gym": False}


def test_vec_env_derives_from_stable_baselines_vecenv_when_asked():
    code = r"""
import json, inspect, numpy as np
from stable_baselines.common.vec_env import VecEnv, VecEnvWrapper
import gym
from basilisk_env_amd._lib import GRAV_PM
from basilisk_env_amd.envs import LeoPowerAttVecEnv
from _oracle_backend import OraclePropagator

out = {"abstract_left": sorted(getattr(LeoPowerAttVecEnv, "__abstractmethods__", ()))}
env = LeoPowerAttVecEnv(6, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, seed=5, propagator_factory=OraclePropagator)
out["isinstance"] = isinstance(env, VecEnv)
out["mro"] = [c.__name__ for c in type(env).__mro__]
out["spaces_ok"] = type(env.observation_space) is gym.spaces.Box and type(env.action_space) is gym.spaces.Discrete and env.num_envs == 6
# every abstract method of the base is the env's own implementation, with a compatible signature
own = {}
for name in sorted(VecEnv.__abstractmethods__):
    f = getattr(LeoPowerAttVecEnv, name)
    own[name] = f.__qualname__.startswith("LeoPowerAttVecEnv.")
out["own"] = own
# ... and a stable-baselines wrapper drives it through the base-class protocol (VecEnv.step = step_async + step_wait)
class W(VecEnvWrapper):
    pass
w = W(env)
ob = w.reset()
obs, rews, dones, infos = w.step(np.zeros(6, int))
out["shapes"] = [list(ob.shape), list(obs.shape), list(rews.shape), list(dones.shape), len(infos)]
out["attr"] = w.get_attr("max_length", [0, 5]); out["seed"] = w.seed(3); out["wrapped"] = w.env_is_wrapped(W)
w.close()
print(json.dumps(out))
"""
    out = _run(code, extra_env={"BSKGPU_SB_VECENV": "1"}, fake_sb=True)
    assert out["abstract_left"] == [] and out["isinstance"] is True and out["spaces_ok"]
    assert out["mro"] == ["LeoPowerAttVecEnv", "VecEnv", "ABC", "object"]
    assert out["own"] and all(out["own"].values()), out["own"]
    assert out["shapes"] == [[6, 5, 1], [6, 5, 1], [6], [6], 6]
    assert out["attr"] == [540, 540] and out["seed"] == [3] * 6 and out["wrapped"] == [False] * 6


def test_vec_env_stays_a_plain_class_unless_stable_baselines_is_in_use():
    """Importing the package never pulls stable-baselines in by itself: without BSKGPU_SB_VECENV=1 and without the caller having
    imported it, the base is ``object`` even when the package is importable; once the caller HAS imported it, it is used."""
    code = r"""
import json, sys
%s
from basilisk_env_amd.envs import LeoPowerAttVecEnv
print(json.dumps({"mro": [c.__name__ for c in LeoPowerAttVecEnv.__mro__], "sb_loaded": "stable_baselines.common.vec_env" in sys.modules}))
"""
    out = _run(code % "", fake_sb=True)
    assert out == {"mro": ["LeoPowerAttVecEnv", "object"], "sb_loaded": False}
    out = _run(code % "import stable_baselines.common.vec_env", fake_sb=True)
    assert out["mro"] == ["LeoPowerAttVecEnv", "VecEnv", "ABC", "object"] and out["sb_loaded"]
