"""CPU: host logic of the gym surface (simulator mirror, leoPowerAttEnv, LeoPowerAttVecEnv) driven
by the oracle-backed propagator stand-in (tests/_oracle_backend.py).  Mirrors the semantics
table of reference envs/leoPowerAttitudeEnvironment.py:65-216 branch by branch."""
import numpy as np
import pytest

from _oracle_backend import OraclePropagator
from basilisk_env_amd import spaces
from basilisk_env_amd._lib import GRAV_PM, GRAV_PM_J2
from basilisk_env_amd.envs import LeoPowerAttVecEnv, leoPowerAttEnv
from basilisk_env_amd.simulators.leoPowerAttitudeSimulator import LEOPowerAttitudeSimulator

KW = {"propagator_factory": OraclePropagator}


def make_env(**extra):
    kw = dict(KW)
    kw.update(extra)
    return leoPowerAttEnv(simulator_kwargs=kw)


def test_constructor_surface():
    env = make_env()
    assert env.max_length == 540 and env.step_duration == 180. and env.failure_penalty == 1
    assert abs(env.wheel_limit - 3000 * 2 * np.pi / 60) < 1e-12 and env.power_max == 20.0
    assert abs(env.reward_mult - 1 / 540) < 1e-18
    assert env.observation_space.shape == (5, 1) and env.action_space.n == 3
    assert isinstance(env, spaces.Env)
    with pytest.raises(RuntimeError):
        env.step(0)      # reset() is mandatory, as in the reference (its lazy path raises TypeError)


def test_reset_observation_quirks():
    """reset(): (5,1) float64; obs[0] is |sigma_BN|, obs[2] is the wheel-speed norm in RPM divided
    by the rad/s limit, obs[3] the charge in W h over power_max — exactly the reference's reset
    observation (…Simulator.py:348-351, …Environment.py:188-191)."""
    env = make_env()
    env.seed(5)
    ob = env.reset()
    ic = env.simulator.initial_conditions
    assert ob.shape == (5, 1) and ob.dtype == np.float64
    assert ob[0, 0] == np.linalg.norm(ic["sigma_init"])
    assert ob[1, 0] == np.linalg.norm(ic["omega_init"])
    assert ob[2, 0] == np.linalg.norm(ic["wheelSpeeds"]) / env.wheel_limit
    assert ob[3, 0] == ic["storedCharge_Init"] / 3600.0 / env.power_max
    assert ob[4, 0] == 0.0
    assert set(ic) >= {"mass", "oe", "rN", "vN", "width", "depth", "height", "sigma_init", "omega_init",
                       "disturbance_magnitude", "disturbance_vector", "wheelSpeeds", "nHat_B", "panelArea",
                       "panelEfficiency", "powerDraw", "storageCapacity", "storedCharge_Init", "sigma_R0N",
                       "controlAxes_B", "K", "Ki", "P", "hs_min", "thrForceSign", "maxCounterValue", "thrMinFireTime"}


def test_seeded_reset_is_reproducible_and_stream_aligned():
    env = make_env()
    env.seed(11)
    a = env.reset()
    b = env.reset()          # second episode continues the same legacy RNG stream
    env.seed(11)
    a2 = env.reset()
    b2 = env.reset()
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and not np.array_equal(a, b)


def test_step_tuple_and_reward():
    env = make_env()
    env.seed(1)
    env.reset()
    ob, reward, done, info = env.step(0)
    assert ob.shape == (5, 1) and isinstance(done, bool) and set(info) == {"full_states", "obs"}
    assert info["full_states"] == [] and info["obs"] is ob
    sim_obs0 = ob[0, 0]
    assert abs(reward - env.reward_mult / (1 + sim_obs0 ** 2)) < 1e-18
    ob, reward, done, info = env.step(1)
    assert reward == 0
    ob, reward, done, info = env.step(2)
    assert reward == 0 and env.curr_step == 3
    assert env.action_episode_memory[-1] == [0, 1, 2]
    with pytest.raises(ValueError):
        env.step(3)


def test_matches_oracle_trajectory():
    """The env's observations equal a direct oracle run on the same ICs (device-side obs/reward
    path vs the host-side formulas of the mirror)."""
    from basilisk_env_amd.simulators.leoPowerAttitudeSimulator import ic_dict_to_block
    from oracle import oracle
    env = make_env()
    env.seed(3)
    env.reset()
    sim = env.simulator
    st = ic_dict_to_block(sim.initial_conditions, 3)
    steps, ticks = np.zeros(1, np.int32), np.zeros(1, np.int32)
    for a in (0, 0, 1, 0):
        ob, reward, done, info = env.step(a)
        o, r, d, w = oracle.step(sim.cfg, st, steps, ticks, [a], 1800)
        assert abs(ob[0, 0] - o[0, 0]) < 1e-15 and abs(ob[1, 0] - o[1, 0]) < 1e-15
        assert abs(ob[2, 0] - o[2, 0]) < 1e-14 and abs(ob[3, 0] - o[3, 0]) < 1e-14
        assert abs(reward - r[0]) < 1e-16 and done == bool(d[0])


def test_episode_length_termination():
    env = make_env()
    env.seed(2)
    env.reset()
    env.max_length = 3
    dones = []
    for _ in range(4):
        ob, r, done, info = env.step(1)
        dones.append(done)
    assert dones == [False, False, False, True]       # curr_step >= max_length checked before the step
    assert info["episode"]["l"] == 3 and "r" in info["episode"]


def test_wheel_overspeed_termination_and_penalty():
    env = make_env()
    env.seed(4)
    env.reset()
    ic = dict(env.simulator.initial_conditions)
    ic["wheelSpeeds"] = np.array([2900.0, 2900.0, 2900.0])     # RPM, norm > 3000
    env.simulator = None
    env.simulator = LEOPowerAttitudeSimulator(.1, 1.0, 180., ic, **KW)
    ob, reward, done, info = env.step(1)
    assert done and ob[2, 0] > 1 and reward == -1 and info["episode"]["r"] == -1


def test_battery_empty_termination():
    env = make_env()
    env.seed(4)
    env.reset()
    ic = dict(env.simulator.initial_conditions)
    ic["storedCharge_Init"] = 0.0
    ic["panelEfficiency"] = 0.0          # dead panel: the -5 W sink keeps the battery at the lower clamp
    env.simulator = LEOPowerAttitudeSimulator(.1, 1.0, 180., ic, **KW)
    ob, reward, done, info = env.step(0)
    assert done and ob[3, 0] == 0
    assert abs(reward - (env.reward_mult / (1 + ob[0, 0] ** 2) - 1)) < 1e-15


def test_reset_init_replays_initial_conditions():
    env = make_env()
    env.seed(9)
    first = env.reset()
    traj = [env.step(0)[0].copy() for _ in range(2)]
    again = env.reset_init()
    assert np.array_equal(first, again)
    traj2 = [env.step(0)[0].copy() for _ in range(2)]
    assert all(np.array_equal(a, b) for a, b in zip(traj, traj2))


def test_simulator_mirror_surface():
    sim = LEOPowerAttitudeSimulator(.1, 1.0, 60., **KW)
    assert sim.dynRate == .1 and sim.fswRate == 1.0 and sim.step_duration == 60. and sim.substeps == 600
    assert sim.obs.shape == (5, 1) and sim.mass == 330 and sim.powerDraw == -5.0
    obs, states, over = sim.run_sim(0)
    assert obs.shape == (5, 1) and states == [] and over is False and sim.simTime == 60.
    assert obs[2, 0] > 1.0          # raw rad/s, not normalised (the env divides)
    sim.close_gracefully()
    with pytest.raises(ValueError):
        LEOPowerAttitudeSimulator(.1, 0.25, 60., **KW)


def test_vec_env_surface_and_autoreset():
    n = 70
    env = LeoPowerAttVecEnv(n, n_rw=4, gravity_model=GRAV_PM_J2, step_duration=2.0, seed=0, **KW)
    assert env.num_envs == n and env.substeps == 20
    ob = env.reset()
    assert ob.shape == (n, 5, 1)
    env.cfg.max_length = 2
    env.propagator.cfg.max_length = 2
    acts = np.zeros(n, np.int64)
    o1, r1, d1, i1 = env.step(acts)
    assert o1.shape == (n, 5, 1) and r1.shape == (n,) and d1.dtype == bool and len(i1) == n and not d1.any()
    env.step(acts)
    ic_before = env._ic.copy()
    o3, r3, d3, i3 = env.step(acts)                    # third step: steps >= max_length -> done
    assert d3.all()
    assert all("episode" in i and "terminal_observation" in i and i["done_reason"]["length"] for i in i3)
    assert i3[0]["episode"]["l"] == 2
    assert not np.array_equal(env._ic, ic_before)      # fresh ICs were drawn
    steps, ticks = env.propagator.get_counters()
    assert (steps == 0).all() and (ticks == 0).all()
    assert np.array_equal(o3[:, 0, 0], np.linalg.norm(env._ic[6:9], axis=0))     # obs of the new episode
    o4, r4, d4, _ = env.step(acts)
    assert not d4.any()
    rsum, ndone = env.batch_stats()
    assert abs(rsum - r4.sum()) < 1e-12 and ndone == 0
    assert env.get_attr("max_length") == [540] * n and env.env_is_wrapped(None) == [False] * n
    with pytest.raises(ValueError):
        env.step(np.full(n, 3))
    env.close()


def test_vec_env_matches_single_env_semantics():
    """The device-side reward/done of the vec env equal the single env's host-side logic."""
    env = make_env(n_rw=3, gravity_model=GRAV_PM)
    env.seed(8)
    env.reset()
    from basilisk_env_amd.simulators.leoPowerAttitudeSimulator import ic_dict_to_block
    ic = ic_dict_to_block(env.simulator.initial_conditions, 3)
    venv = LeoPowerAttVecEnv(1, n_rw=3, gravity_model=GRAV_PM, **KW)
    venv.reset(ic)
    for a in (0, 1, 0):
        ob, rew, done, info = env.step(a)
        vob, vrew, vdone, _ = venv.step([a])
        assert np.abs(vob[0] - ob).max() < 1e-14 and abs(vrew[0] - rew) < 1e-16 and bool(vdone[0]) == done


def test_vec_env_device_reset_bookkeeping():
    """device_reset_pool: finished envs restart from the staged pool (rule of include/bskgpu.h), the
    step returns the new episode's first observation and the old one as terminal_observation."""
    n = 40
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, seed=1, device_reset_pool=16, **KW)
    env.cfg.max_length = 2
    env.propagator.cfg.max_length = 2
    env.reset()
    pool = env.propagator._pool
    acts = np.zeros(n, np.int64)
    env.step(acts)
    env.step(acts)
    obs, rew, done, infos = env.step(acts)
    assert done.all() and all("terminal_observation" in i and i["episode"]["l"] == 2 for i in infos)
    st = env.propagator.get_state()
    for i in range(n):
        slot = ((i * 2654435761 + 0 * 40503 + 12345) & 0xFFFFFFFF) % 16
        assert np.array_equal(st[:, i], pool[:, slot])
        assert obs[i, 0, 0] == np.linalg.norm(pool[6:9, slot]) and obs[i, 4, 0] == 1.0
    steps, ticks = env.propagator.get_counters()
    assert (steps == 0).all() and (ticks == 0).all()
    _, eps = env.propagator.get_terminal_obs()
    assert (eps == 1).all()
    o2, r2, d2, _ = env.step(acts)
    assert not d2.any()
    env.close()


# ------------------------------------------------------------------------------------------------------------
# stable-baselines drops in unchanged (reference README.md:13, basilisk_env/__init__.py:6-9): the access pattern
# of its VecEnv wrappers, restated minimally (stable-baselines is not installed here), drives the batched env.
class _FakeVecMonitor(object):
    """What VecMonitor does to a VecEnv: accumulates returns, and WRITES 'episode' into infos[i] when done."""

    def __init__(self, venv):
        self.venv, self.returns, self.lengths = venv, np.zeros(venv.num_envs), np.zeros(venv.num_envs, int)

    def reset(self):
        self.returns[:] = 0
        self.lengths[:] = 0
        return self.venv.reset()

    def step(self, actions):
        self.venv.step_async(actions)
        obs, rews, dones, infos = self.venv.step_wait()
        self.returns += rews
        self.lengths += 1
        infos = list(infos)
        for i in range(len(dones)):
            if dones[i]:
                info = infos[i].copy()
                info["episode"] = {"r": self.returns[i], "l": self.lengths[i], "t": 0.0}
                infos[i] = info
                self.returns[i] = 0
                self.lengths[i] = 0
            else:
                infos[i]["monitor_seen"] = infos[i].get("monitor_seen", 0) + 1     # a wrapper writing into info
        return obs, rews, dones, infos


class _FakeVecNormalize(object):
    """What VecNormalize does: running mean/var over obs batches, and it normalises infos[i]['terminal_observation']."""

    def __init__(self, venv):
        self.venv, self.count, self.mean = venv, 0, 0.0

    def _upd(self, obs):
        self.count += obs.shape[0]
        self.mean = self.mean + (obs.mean(axis=0) - self.mean) * obs.shape[0] / self.count

    def reset(self):
        obs = self.venv.reset()
        self._upd(obs)
        return obs - self.mean

    def step(self, actions):
        obs, rews, dones, infos = self.venv.step(actions)
        self._upd(obs)
        for i in np.flatnonzero(dones):
            assert infos[i]["terminal_observation"].shape == obs[i].shape
            infos[i]["terminal_observation"] = infos[i]["terminal_observation"] - self.mean
        return obs - self.mean, rews, dones, infos


@pytest.mark.parametrize("device_pool", [0, 8])
def test_vec_env_under_stable_baselines_style_wrappers(device_pool):
    n = 12
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, seed=5, device_reset_pool=device_pool, **KW)
    env.cfg.max_length = 2
    env.propagator.cfg.max_length = 2
    w = _FakeVecNormalize(_FakeVecMonitor(env))
    ob = w.reset()
    assert ob.shape == (n, 5, 1)
    seen_done = 0
    for k in range(7):
        obs, rews, dones, infos = w.step(np.zeros(n, int))
        assert len(infos) == n and len({id(i) for i in infos}) == n          # one dict per env, none shared
        for i in range(n):
            if dones[i]:
                seen_done += 1
                assert infos[i]["episode"]["l"] == 3 and "terminal_observation" in infos[i]
                assert "monitor_seen" not in infos[i]
            else:
                assert infos[i] == {"monitor_seen": 1}                       # nothing leaked from earlier steps / envs
    assert seen_done == 2 * n
    env.close()


def test_vec_env_protocol_details():
    n = 5
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, seed=5, **KW)
    env.reset()
    with pytest.raises(RuntimeError):
        env.step_wait()                                  # no pending step_async
    env.step_async(np.zeros(n, int))
    env.step_wait()
    with pytest.raises(RuntimeError):
        env.step_wait()                                  # consumed
    assert env.get_attr("max_length", indices=[0, 3]) == [540, 540]
    assert env.get_attr("max_length", indices=2) == [540]
    assert env.env_method("batch_stats", indices=[1, 2, 4])[0] == env.batch_stats()
    assert len(env.env_method("batch_stats", indices=[1, 2, 4])) == 3 and len(env.env_method("batch_stats")) == n
    assert env.env_is_wrapped(object, indices=[0]) == [False]
    env.set_attr("reward_mult", 0.5)
    assert env.get_attr("reward_mult") == [0.5] * n
    with pytest.raises(ValueError):
        env.set_attr("reward_mult", 0.1, indices=[0])
    with pytest.raises(IndexError):
        env.get_attr("max_length", indices=[n])
    assert env.seed(3) == [3] * n
    env.close()


@pytest.mark.parametrize("device_sampler", [False, True])
def test_vec_env_reset_init_after_device_reset(device_sampler):
    """reset_init() replays the CURRENT episodes' initial conditions, also for envs the device restarted itself."""
    n = 9
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, seed=2, device_reset_pool=4,
                            device_sampler=device_sampler, **KW)
    env.cfg.max_length = 1
    env.propagator.cfg.max_length = 1
    env.reset()
    acts = np.zeros(n, int)
    env.step(acts)
    _, _, done, _ = env.step(acts)                      # every env finishes and is restarted on the "device"
    assert done.all()
    restarted = env.propagator.get_state()
    assert np.array_equal(env._ic, restarted)
    env.step(acts)
    first = env.reset_init()
    assert np.array_equal(env.propagator.get_state(), restarted)
    assert np.array_equal(first[:, 0, 0], np.linalg.norm(restarted[6:9], axis=0))
    env.close()


def test_single_env_reuses_its_device_handle_across_resets():
    """The reference rebuilds its simulator on every reset (:184-185); here the propagator handle behind it is
    parked and taken over by the next simulator: one construction for any number of resets."""
    from basilisk_env_amd.simulators import leoPowerAttitudeSimulator as S
    S.drain_idle_propagators()
    made = []

    def factory(cfg, n_envs, device=0):
        made.append(1)
        return OraclePropagator(cfg, n_envs, device=device)

    env = leoPowerAttEnv(simulator_kwargs={"propagator_factory": factory})
    env.seed(1)
    a = env.reset()
    env.step(0)
    env.reset()
    env.step(1)
    b = env.reset_init()
    assert len(made) == 1
    env.seed(1)
    a2 = env.reset()
    assert np.array_equal(a, a2) and len(made) == 1          # a re-used handle starts from a clean slate
    steps, ticks = env.simulator.propagator.get_counters()
    assert steps[0] == 0 and ticks[0] == 0
    env.close()
    assert not S._IDLE_PROPAGATORS


def test_info_list_is_a_list_of_distinct_dicts_made_on_demand():
    """leoPowerAttitudeVecEnv.InfoList: what step_wait returns as ``infos``.  It must behave as the plain list of one
    dict per env that stable-baselines' wrappers index, slice, iterate, copy and write into, without building the
    dicts nobody looks at (65 536 of them cost more than the step kernel)."""
    import copy
    import pickle
    from basilisk_env_amd.envs.leoPowerAttitudeVecEnv import InfoList
    infos = InfoList(5)
    infos[3] = {"episode": {"r": 1.0, "l": 2}}
    assert isinstance(infos, list) and len(infos) == 5
    assert infos[0] == {} and infos[0] is infos[0] and infos[0] is not infos[1]      # made once, never shared
    infos[1]["x"] = 1                                                               # a wrapper writing into info
    assert infos[1] == {"x": 1} and infos[2] == {} and infos[-1] == {}
    assert [i.get("episode") for i in infos] == [None, None, None, {"r": 1.0, "l": 2}, None]
    assert infos[1:4] == [{"x": 1}, {}, {"episode": {"r": 1.0, "l": 2}}] and type(infos[:]) is list
    assert list(infos)[1] is infos[1] and list(reversed(infos))[0] is infos[4]
    assert infos == [{}, {"x": 1}, {}, {"episode": {"r": 1.0, "l": 2}}, {}] and {"x": 1} in infos
    for clone in (copy.copy(infos), copy.deepcopy(infos), pickle.loads(pickle.dumps(infos)), infos.copy(), infos + []):
        assert type(clone) is list and clone == list(infos)
    assert len({id(d) for d in infos}) == 5
    with pytest.raises(TypeError):
        infos * 2
    assert InfoList(1)[0] == {} and len(InfoList(0)) == 0
    # untouched slots cost nothing: no dict exists until somebody asks
    big = InfoList(100000)
    assert list.__getitem__(big, 99999) is None and big[99999] == {} and list.__getitem__(big, 99999) is big[99999]


def test_step_wait_has_no_stall_when_every_episode_ends_together():
    """With the reference's ``max_length`` and a common ``reset()`` ALL envs of a batch finish on the same step
    (envs/leoPowerAttitudeEnvironment.py:98-99: every 541st step).  That step's host work must be array work: 65 536 envs,
    ``max_length = 2``, the oracle-backed engine (step_wait itself never calls into it for arithmetic).  Before round 5 the
    all-done step_wait built 65 536 dicts + (5,1) arrays + one pool_slot() call per env: 278 ms with the device pool, 425 ms
    with host resets, against 1.2 ms for an ordinary step."""
    import time
    n = 65536
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=0.1, seed=3, device_reset_pool=1024, **KW)
    env.cfg.max_length = 2
    env.propagator.cfg.max_length = 2
    env.reset()
    acts = np.zeros(n, np.int64)
    ordinary, all_done = [], []
    for k in range(9):
        env.step_async(acts)
        t0 = time.perf_counter()
        obs, rew, done, infos = env.step_wait()
        dt = time.perf_counter() - t0
        (all_done if done.all() else ordinary).append(dt)
        assert done.all() == (k % 3 == 2) and (done.all() or not done.any())
    assert len(all_done) == 3
    assert min(all_done) < 5e-3, (all_done, ordinary)                  # 278 ms before
    assert min(all_done) < 4.0 * min(ordinary) + 2e-3, (all_done, ordinary)
    # ... and the lazily built infos are the contract's: episode / terminal_observation / done_reason per finished env
    assert isinstance(infos, list) and len(infos) == n
    term, eps = env.propagator.get_terminal_obs()
    for i in (0, 1, 777, n - 1):
        assert infos[i]["episode"] == {"r": pytest.approx(float(3 * rew[i]), rel=1e-6), "l": 2} or infos[i]["episode"]["l"] == 2
        assert infos[i]["done_reason"] == {"length": True, "wheels": False, "battery": False, "orbit": False}
        assert infos[i]["terminal_observation"].shape == (5, 1) and np.array_equal(infos[i]["terminal_observation"][:, 0], term[:, i])
        assert infos[i] is infos[i]
    assert (eps == 3).all()
    # reset_init() after three device-side restarts of every env: the pool columns of the slot rule, gathered on demand
    from basilisk_env_amd.envs.leoPowerAttitudeVecEnv import pool_slot
    ic = env._ic
    for i in (0, 5, n - 1):
        assert np.array_equal(ic[:, i], env._pool[:, pool_slot(i, 2, 1024)])
    env.close()


def test_all_done_step_with_host_resets_is_array_work_too():
    """The same step without a device pool: fresh initial conditions for the whole batch are sampled on the host (26 ms for
    65 536 - the floor of this path) and uploaded without a mask; nothing else per env."""
    import time
    n = 65536
    env = LeoPowerAttVecEnv(n, n_rw=3, gravity_model=GRAV_PM, step_duration=0.1, seed=3, **KW)
    env.cfg.max_length = 1
    env.propagator.cfg.max_length = 1
    env.reset()
    acts = np.zeros(n, np.int64)
    env.step(acts)
    best = 1e9
    for _ in range(2):
        env.step_async(acts)
        t0 = time.perf_counter()
        obs, rew, done, infos = env.step_wait()
        best = min(best, time.perf_counter() - t0)
        assert done.all()
        env.step(acts)
    assert best < 0.12, best                                           # 425 ms before
    assert np.array_equal(obs[:, 0, 0], np.linalg.norm(env._ic[6:9], axis=0))      # the new episodes' first observations
    assert infos[4242]["episode"]["l"] == 1 and "terminal_observation" in infos[4242]
    env.close()


@pytest.mark.parametrize("pool", [0, 8])
@pytest.mark.parametrize("const", [None, 0])
def test_rollout_is_n_steps_of_step_including_the_bookkeeping(pool, const):
    """LeoPowerAttVecEnv.rollout (the reference's mains for a whole batch, envs/leoPowerAttitudeEnvironment.py:218-231): rows = what
    step() number t returns, and episode returns / lengths / the running episodes' initial conditions end as after those steps."""
    n, T = 12, 7
    kw = dict(KW, n_rw=3, step_duration=1.0, seed=3, device_reset_pool=pool, auto_reset=bool(pool))
    probe = LeoPowerAttVecEnv(n, **kw)
    cfg = probe.cfg
    probe.close()
    cfg.max_length = 2
    a, b = (LeoPowerAttVecEnv(n, cfg=cfg, step_duration=1.0, seed=3, device_reset_pool=pool, auto_reset=bool(pool), propagator_factory=OraclePropagator)
            for _ in range(2))
    a.reset(); b.reset()
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 3, (T, n))
    rows = [a.step(acts[t] if const is None else np.full(n, const)) for t in range(T)]
    obs, rew, dones, why = b.rollout(T, actions=acts if const is None else None, constant_action=const or 0)
    assert obs.shape == (T, n, 5, 1) and dones.any() and not dones.all()
    for t in range(T):
        assert np.array_equal(obs[t], rows[t][0]) and np.array_equal(rew[t], rows[t][1]) and np.array_equal(dones[t], rows[t][2])
    assert np.array_equal(a.episode_returns, b.episode_returns) and np.array_equal(a.episode_lengths, b.episode_lengths)
    if pool:
        assert np.array_equal(a._ic, b._ic)
    with pytest.raises(ValueError):
        b.rollout(2, constant_action=5)
    host = LeoPowerAttVecEnv(n, n_rw=3, step_duration=1.0, propagator_factory=OraclePropagator)      # host-side auto-reset
    host.reset()
    with pytest.raises(ValueError):
        host.rollout(2)
    for e in (a, b, host):
        e.close()


def test_demo_batch_runs_whole_episodes_in_one_call_each(capsys):
    from basilisk_env_amd.envs.leoPowerAttitudeEnvironment import demo_batch
    out = demo_batch(num_envs=5, episodes=2, n_rw=3, step_duration=1.0, propagator_factory=OraclePropagator, power=False)
    assert len(out) == 2
    obs, ret, length = out[0]
    assert obs.shape[1:] == (5, 5) and ret.shape == (5,) and np.all(length >= 1) and np.all(ret > 0.0)       # (action 0 earns the nadir reward)
    assert capsys.readouterr().out.count("episode") == 2


def test_demo_main_rolls_whole_episodes_of_one_action(capsys):
    """What the reference module does when run as a script (envs/leoPowerAttitudeEnvironment.py:218-231: make the env, reset, seed,
    step action 0 until the episode is over, keep the observation history) - `demo()`, here on the oracle-backed engine; with gym
    absent `make_env` builds the class itself (tests/test_gym_boundary.py runs the gym.make branch)."""
    from basilisk_env_amd.envs.leoPowerAttitudeEnvironment import demo
    hists = demo(episodes=1, env_kwargs={"simulator_kwargs": dict(KW)})
    assert len(hists) == 1 and hists[0].shape[0] == 5 and 1 <= hists[0].shape[1] <= 541
    assert np.isfinite(hists[0]).all() and np.all(hists[0][4] >= 0.0) and np.all(hists[0][4] <= 1.0)      # obs[4]: sunlit fraction
    assert capsys.readouterr().out.count("episode 0:") == 1


def test_simulator_module_main_runs_360_steps_of_action_0(capsys):
    """The reference simulator module's own main (simulators/leoPowerAttitudeSimulator.py:657-694): 60 s steps, 360 calls of run_sim(0),
    the five observation entries kept - `simulators.leoPowerAttitudeSimulator.demo()`, here shortened and on the oracle-backed engine."""
    from basilisk_env_amd.simulators.leoPowerAttitudeSimulator import create_leoPowerAttSimulator, demo
    obs = demo(steps=12, **KW)
    assert obs.shape == (12, 5) and np.isfinite(obs).all() and np.all(obs[:, 4] >= 0.0) and np.all(obs[:, 4] <= 1.0)
    assert "12 steps of 60 s under action 0" in capsys.readouterr().out
    assert callable(create_leoPowerAttSimulator)
