"""GPU: the device-resident env surface (row f4's purpose: the training loop stays on the GPU) and the one-process
multi-handle batch — both must give EXACTLY what the host path / the unsharded batch give.

* ``LeoPowerAttVecEnv.step_tensors``: device int32 actions in, device obs / reward / done out, no host sync;
  compared bit for bit with ``step()`` of a twin env through episode ends and device-side resets, on the handle's
  own stream (cross-stream waits) and on torch's stream (no ordering needed).
* DLPack / ``__cuda_array_interface__`` views alias the library's buffers.
* ``ShardedVecEnv(devices=[0, 0])``: two handles and streams on the one card == the unsharded env, bit for bit;
  pinned per-device D2H; the direct gather's own-shard path.
"""
import numpy as np
import pytest

from basilisk_env_amd._lib import FLAG_AUTO_RESET, GRAV_PM, GRAV_PM_J2
from basilisk_env_amd.envs import LeoPowerAttVecEnv
from basilisk_env_amd.sharded import ShardedPropagator, ShardedVecEnv
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

pytestmark = pytest.mark.gpu


def _twin_envs(n, own_stream, **extra):
    import torch
    kw = dict(n_rw=3, gravity_model=GRAV_PM_J2, step_duration=2.0, seed=7, device_reset_pool=32)
    kw.update(extra)
    probe = LeoPowerAttVecEnv(n, **kw)
    cfg = probe.cfg
    probe.close()
    cfg.max_length = 3
    host = LeoPowerAttVecEnv(n, cfg=cfg, step_duration=2.0, seed=7, device_reset_pool=32)
    stream = None if own_stream else torch.cuda.current_stream().cuda_stream
    dev = LeoPowerAttVecEnv(n, cfg=cfg, step_duration=2.0, seed=7, device_reset_pool=32, stream=stream)
    return host, dev


@pytest.mark.parametrize("own_stream", [True, False])
@pytest.mark.parametrize("n", [64, 333])
def test_step_tensors_is_bit_identical_to_host_path(n, own_stream):
    import torch
    host, dev = _twin_envs(n, own_stream)
    ob_h = host.reset()
    ob_d = dev.reset_tensors()
    # (the host path computes the reset observation with numpy, the device path reads the reset kernel's own: one ulp apart at most)
    assert ob_d.is_cuda and tuple(ob_d.shape) == (n, 5, 1) and np.allclose(ob_d.cpu().numpy(), ob_h, rtol=4e-16, atol=0)
    rng = np.random.default_rng(3)
    n_done = 0
    for step in range(8):
        a = rng.integers(0, 3, n).astype(np.int32)
        oh, rh, dh, ih = host.step(a)
        at = torch.as_tensor(a, device="cuda")
        od, rd, dd, info = dev.step_tensors(at)
        assert od.is_cuda and rd.is_cuda and dd.is_cuda and dd.dtype == torch.bool and tuple(od.shape) == (n, 5, 1)
        # consumers on torch's current stream are ordered after the kernel: no explicit synchronisation here
        assert np.array_equal(od.cpu().numpy(), oh)
        assert np.array_equal(rd.cpu().numpy(), rh)
        assert np.array_equal(dd.cpu().numpy(), dh)
        if dh.any():
            term = info["terminal_observation"].cpu().numpy()
            for i in np.flatnonzero(dh):
                assert np.array_equal(term[i], ih[i]["terminal_observation"])
            n_done += int(dh.sum())
        assert np.array_equal(info["episodes"].cpu().numpy(), host.propagator.get_terminal_obs()[1])
    assert n_done >= n                                   # every env went through a device-side reset
    assert np.array_equal(dev.get_state(), host.get_state())
    host.close()
    dev.close()


def test_step_tensors_argument_checks():
    import torch
    env = LeoPowerAttVecEnv(64, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, power=False)
    env.reset()
    with pytest.raises(ValueError):                       # host-side auto-reset cannot serve the tensor path
        env.step_tensors(torch.zeros(64, dtype=torch.int32, device="cuda"))
    env.close()
    env = LeoPowerAttVecEnv(64, n_rw=3, gravity_model=GRAV_PM, step_duration=1.0, power=False, auto_reset=False)
    env.reset()
    for bad in (torch.zeros(64, dtype=torch.int16, device="cuda"), torch.zeros(63, dtype=torch.int32, device="cuda"),
                torch.zeros(64, dtype=torch.int32), np.zeros(64, np.int32), torch.zeros(128, dtype=torch.int64, device="cuda")[::2]):
        with pytest.raises(ValueError):
            env.step_tensors(bad)
    ob, rew, done, info = env.step_tensors(torch.ones(64, dtype=torch.int32, device="cuda"))
    assert "terminal_observation" not in info and not bool(done.any())
    env.close()


@pytest.mark.parametrize("n", [64, 200])
def test_device_episode_statistics_equal_the_host_paths(n):
    """Row f4's bookkeeping on the device: info['episode'] = {'r', 'l'} of the host path (the Monitor convention of the
    reference, envs/leoPowerAttitudeEnvironment.py:130-135) against the kernel's term_return / term_len, and the running
    returns, bit for bit through more than n device-side resets; int32 and int64 action tensors alternate."""
    import torch
    host, dev = _twin_envs(n, own_stream=False)
    host.reset()
    dev.reset_tensors()
    rng = np.random.default_rng(11)
    n_done = 0
    for step in range(9):
        a = rng.integers(0, 3, n)
        oh, rh, dh, ih = host.step(a)
        at = torch.as_tensor(a.astype(np.int64 if step % 2 else np.int32), device="cuda")
        od, rd, dd, info = dev.step_tensors(at)
        assert np.array_equal(rd.cpu().numpy(), rh) and np.array_equal(dd.cpu().numpy(), dh)
        er, el = info["episode_r"].cpu().numpy(), info["episode_l"].cpu().numpy()
        for i in np.flatnonzero(dh):
            assert er[i] == ih[i]["episode"]["r"] and int(el[i]) == ih[i]["episode"]["l"]
        assert np.array_equal(info["episode_return"].cpu().numpy(), host.episode_returns)
        n_done += int(dh.sum())
    assert n_done >= n
    host.close()
    dev.close()


def test_step_tensors_outputs_are_the_kernels_own_buffers():
    """step_tensors launches no torch kernel of its own: observation = the row-major buffer the kernel writes (contiguous, a
    policy's reshape is a view), done = a bool view of the kernel's 0 / 1 byte, everything aliases library memory."""
    import torch
    n = 300
    host, dev = _twin_envs(n, own_stream=False)
    dev.reset_tensors()
    v = dev.propagator.device_views()
    ptr = lambda k: v[k].__cuda_array_interface__["data"][0]
    od, rd, dd, info = dev.step_tensors(torch.zeros(n, dtype=torch.int64, device="cuda"))
    assert od.is_contiguous() and od.data_ptr() == ptr("obs_rowmajor") and od.reshape(n, 5).data_ptr() == od.data_ptr()
    assert dd.dtype == torch.bool and dd.data_ptr() == ptr("done") and rd.data_ptr() == ptr("reward")
    assert info["episode_r"].data_ptr() == ptr("terminal_return") and info["episode_l"].dtype == torch.int32
    soa = torch.as_tensor(v["obs"], device="cuda")
    assert torch.equal(od.reshape(n, 5), soa.t())           # the two layouts hold the same numbers
    assert torch.equal(dd, info["reason"].ne(0))
    host.close()
    dev.close()


def test_reset_tensors_and_the_loop_move_nothing_over_pcie():
    """With the pool drawn on the GPU, reset_tensors + step_tensors issue NO host <-> device copy and NO stream
    synchronisation (the library counts its own: bsk_debug_counters), and the reset observation is the reset kernel's
    own output - equal to what the host path computes from the same pool."""
    import torch
    n = 500
    kw = dict(n_rw=4, step_duration=1.0, seed=5, device_reset_pool=64, device_sampler=True)
    env = LeoPowerAttVecEnv(n, stream=torch.cuda.current_stream().cuda_stream, **kw)
    twin = LeoPowerAttVecEnv(n, **kw)
    ob_host = twin.reset()
    torch.cuda.synchronize()
    c0 = BatchedPropagator.debug_counters()
    ob = env.reset_tensors()
    for _ in range(5):
        ob, rew, done, info = env.step_tensors((ob.reshape(n, 5)[:, :3]).argmax(dim=1))      # int64 actions, in place
    c1 = BatchedPropagator.debug_counters()
    assert c1 == c0, (c0, c1)
    ob0 = env.reset_tensors()
    assert ob0.data_ptr() == env._torch_views()["obs_rowmajor"].data_ptr()
    torch.cuda.synchronize()
    # second device-side reset of `env` = second episode's pool slots; the twin's first reset used the first ones: replay
    env2 = LeoPowerAttVecEnv(n, stream=torch.cuda.current_stream().cuda_stream, **kw)
    assert np.allclose(env2.reset_tensors().cpu().numpy(), ob_host, rtol=4e-16, atol=0)
    assert np.array_equal(env2.reset_init(), ob_host)                   # the IC mirror is rebuilt from the slot rule
    for e in (env, twin, env2):
        e.close()


def test_host_resets_leave_the_first_observation_in_the_device_buffers():
    """bsk_reset (all / masked) writes the new episode's first observation and zeroes reward / reason for the envs it restarts."""
    n = 150
    cfg = default_config(4, GRAV_PM_J2)
    p = BatchedPropagator(cfg, n)
    ic = sample_ic_batch(n, 4, seed=3)
    p.reset(ic)
    env = LeoPowerAttVecEnv.__new__(LeoPowerAttVecEnv)
    env.n_rw, env.wheel_limit, env.power_max = 4, cfg.wheel_limit, cfg.power_max
    want = env._initial_obs(ic)
    obs, rew, done, why = p.get_obs()
    assert np.allclose(obs, want, rtol=1e-15, atol=0) and not rew.any() and not why.any()
    p.step(np.zeros(n, np.int32), 4)
    stepped = p.get_obs()[0]
    mask = (np.arange(n) % 4 == 1).astype(np.uint8)
    ic2 = sample_ic_batch(n, 4, seed=4)
    p.reset(ic2, mask)
    obs2, rew2, _, _ = p.get_obs()
    m = mask.astype(bool)
    assert np.allclose(obs2[:, m], env._initial_obs(ic2)[:, m], rtol=1e-15, atol=0) and np.array_equal(obs2[:, ~m], stepped[:, ~m])
    assert not rew2[m].any() and rew2[~m].any()
    p.close()


def test_on_device_policy_loop_runs_without_host_sync():
    """A linear -> argmax policy on the GPU drives the env for 50 steps; the only host read is at the end."""
    import torch
    n = 4096
    env = LeoPowerAttVecEnv(n, n_rw=4, step_duration=1.0, seed=1, device_reset_pool=256, device_sampler=True,
                            stream=torch.cuda.current_stream().cuda_stream)
    ob = env.reset_tensors()
    w = torch.randn(5, 3, dtype=torch.float64, device="cuda")
    ret = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(50):
        act = (ob.reshape(n, 5) @ w).argmax(dim=1).to(torch.int32)
        ob, rew, done, info = env.step_tensors(act)
        ret += rew
    assert bool(torch.isfinite(ret).all()) and bool(torch.isfinite(ob).all())
    env.close()


def test_batch_stats_are_the_last_steps_whatever_was_reset_since():
    """bsk_get_batch_stats* = sum of rewards and number of finished envs of the LAST STEP.  They are formed from the reward
    buffer by a kernel of their own (the step kernel's epilogue carries no reward reduction), so every reset entry point
    takes the snapshot before it zeroes the restarted envs' rewards: the vec env auto-resets before a loop reads them."""
    n = 777
    cfg = default_config(4, GRAV_PM_J2)
    cfg.max_length = 2
    cfg.flags |= FLAG_AUTO_RESET          # (a pool may be staged; with max_length reached the kernel itself restarts envs too)
    p = BatchedPropagator(cfg, n)
    ic = sample_ic_batch(n, 4, seed=11)
    p.reset(ic)
    p.set_ic_pool(sample_ic_batch(64, 4, seed=12))
    rng = np.random.default_rng(2)
    assert p.batch_stats() == (0.0, 0)                       # before any step
    for _ in range(3):
        p.step(rng.integers(0, 3, n).astype(np.int32), 4)
    obs, rew, done, why = p.get_obs()
    want_s, want_d = float(rew.sum()), int((why != 0).sum())
    assert want_d == n and want_s != 0.0                     # (max_length reached: every env finished in the last step)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    p.reset(sample_ic_batch(n, 4, seed=13), mask)            # host ICs, masked: zeroes those envs' rewards
    assert np.all(p.get_obs()[1][mask.astype(bool)] == 0.0)
    s1, d1 = p.batch_stats()
    assert abs(s1 - want_s) < 1e-12 * n and d1 == want_d
    p.reset_from_pool(mask)                                  # pool, masked
    p.reset_from_pool_device(0)                              # pool, all, asynchronous
    p.reset(ic)                                              # everything
    assert np.all(p.get_obs()[1] == 0.0)
    assert p.batch_stats() == (s1, d1)                       # still the last step's, bit for bit
    p.step(np.zeros(n, np.int32), 1)
    obs, rew, done, why = p.get_obs()
    s2, d2 = p.batch_stats()
    assert abs(s2 - float(rew.sum())) < 1e-12 * n and d2 == int((why != 0).sum()) and (s2, d2) != (s1, d1)
    p.close()


def _stats_order(rew):
    """bsk_get_batch_stats' reward sum, operation for operation (stats_kernel): per 64 envs an xor butterfly, wave w into slot
    w mod 256 in ascending order, then a halving tree over the 256 slots."""
    n = len(rew)
    nw = (n + 63) // 64
    v = np.zeros(nw * 64)
    v[:n] = rew
    v = v.reshape(nw, 64)
    idx = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        v = v + v[:, idx ^ off]
    ws = v[:, 0]
    slots = np.zeros(256)
    for w in range(nw):
        slots[w & 255] += ws[w]
    off = 128
    while off:
        slots[:off] += slots[off:2 * off]
        off >>= 1
    return float(slots[0])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20000])
def test_batch_stats_reward_sum_has_a_fixed_order(n):
    """Bitwise reproducible: the sum is the documented tree, whatever the batch size."""
    cfg = default_config(4, GRAV_PM_J2)
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=21))
    rng = np.random.default_rng(3)
    for k in (1, 3):
        p.step(rng.integers(0, 3, n).astype(np.int32), k)
        s, d = p.batch_stats()
        obs, rew, done, why = p.get_obs()
        assert s == _stats_order(rew) and d == int((why != 0).sum())
    p.close()


@pytest.mark.parametrize("n", [65536, 131072 + 77, (1 << 20) + 77, 1 << 22])
def test_batch_stats_order_holds_at_every_scale(n):
    """The two-level stats_kernel (a multi-workgroup first level, the last workgroup joins the wave sums) gives the documented
    tree bit for bit from one workgroup's worth of envs to 4 Mi (2 048 workgroups, eight trips each), the done count included,
    and asking twice - or through the device pointer - changes nothing."""
    import ctypes
    cfg = default_config(4, GRAV_PM_J2)
    cfg.max_length = 1
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=22))
    rng = np.random.default_rng(n)
    for k in range(3):
        p.step(rng.integers(0, 3, n).astype(np.int32), 1)
        s, d = p.batch_stats()
        obs, rew, done, why = p.get_obs()
        assert s == _stats_order(rew) and d == int((why != 0).sum()), (n, k)
        assert (d > 0) == (k >= 1)                                  # episodes of max_length 1 end from the second step on
        assert p.batch_stats() == (s, d)
        ptr = p.batch_stats_device()
        p.sync()
        from basilisk_env_amd import _hip
        host = (ctypes.c_double * 2)()
        _hip.check(_hip.runtime().hipMemcpyAsync(ctypes.cast(host, ctypes.c_void_p), ctypes.c_void_p(ptr), 16, _hip.hipMemcpyDeviceToHost, ctypes.c_void_p(0)), "hipMemcpyAsync")
        _hip.stream_sync(0)
        assert (host[0], host[1]) == (s, float(d))
    p.close()


@pytest.mark.parametrize("case", ["bare-1", "bare-65", "bare-1000", "bare-65536", "bare-1048653", "bare-2500000", "scenario-1000", "general-1000", "pair-2000", "tri-1000",
                                  "sh-300"])
def test_wave_sums_formed_inside_the_step_launch_give_the_same_bits(case):
    """bsk_set_step_stats: the step kernel's epilogue forms the first level of the batch reduction itself (every kernel form: one
    wave, pair, three waves, harmonics pairs, general inertia, the scenario levels) and a request behind it runs the join kernel
    alone (up to 2 Mi spacecraft; above, the two-level form stays).  Same tree, same bits as the two-launch form and as the documented
    order; switching it off, a rollout launch or a masked reset in between fall back / keep the snapshot as before."""
    from basilisk_env_amd._lib import FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_SH
    from helpers import general_hub
    kind, n = case.split("-")
    n = int(n)
    cfg = default_config(4, GRAV_SH if kind == "sh" else GRAV_PM_J2)
    cfg.max_length = 2
    sub = 1
    if kind in ("scenario", "general", "pair", "tri"):
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        cfg.base_density, cfg.scale_height = 1e-9, 100e3
        if kind == "general":
            general_hub(cfg)
    if kind == "sh":
        cfg.sh_degree = 8
    if kind in ("pair", "tri"):
        sub = 64
    import os
    old = {k: os.environ.get(k) for k in ("BSKGPU_PAIR", "BSKGPU_TRI")}
    os.environ["BSKGPU_PAIR"], os.environ["BSKGPU_TRI"] = "1" if kind == "pair" else "0", "1" if kind == "tri" else "0"
    try:
        p = BatchedPropagator(cfg, n)
        os.environ["BSKGPU_PAIR"] = os.environ["BSKGPU_TRI"] = "0"
        twin = BatchedPropagator(cfg, n)                  # (always the single-wave form, sums formed by stats_kernel)
    finally:
        for k_, v_ in old.items():
            if v_ is None:
                del os.environ[k_]
            else:
                os.environ[k_] = v_
    if kind == "sh":
        from basilisk_env_amd.simulators.dynamics.gravity_sh import synthetic_sh_coefficients
        cbar, sbar = synthetic_sh_coefficients(8, seed=3)
        p.set_gravity_sh(8, cbar, sbar)
        twin.set_gravity_sh(8, cbar, sbar)
    ic = sample_ic_batch(n, 4, seed=31)
    p.reset(ic)
    twin.reset(ic)
    p.set_step_stats(True)
    rng = np.random.default_rng(n)
    for k in range(4):
        act = rng.integers(0, 3, n).astype(np.int32)
        if k == 3:
            p.set_step_stats(False)                      # off again: the two-launch form describes this step
        p.step(act, sub)
        twin.step(act, sub)
        if kind in ("pair", "tri"):
            assert kind in p.kernel_info()["name"]
        if kind == "general":
            assert "diag" not in p.kernel_info()["name"]
        s, d = p.batch_stats()
        obs, rew, done, why = p.get_obs()
        assert (s, d) == twin.batch_stats() and s == _stats_order(rew) and d == int((why != 0).sum()), (case, k)
        assert np.array_equal(rew, twin.get_obs()[1])
        assert (d > 0) == (k >= 2)
        if k == 1:                                       # a masked reset zeroes rewards: the snapshot (join alone) came first
            mask = (rng.random(n) < 0.5).astype(np.uint8)
            p.reset(sample_ic_batch(n, 4, seed=32), mask)
            twin.reset(sample_ic_batch(n, 4, seed=32), mask)
            assert p.batch_stats() == (s, d)
    p.close()
    twin.close()


def test_a_rollout_launch_between_steps_does_not_leave_stale_wave_sums():
    """bsk_step_n writes rewards of its last env step but no wave sums: the request behind it must run both levels."""
    n = 3000
    cfg = default_config(4, GRAV_PM_J2)
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=33))
    p.set_step_stats(True)
    p.step(np.zeros(n, np.int32), 1)
    s0 = p.batch_stats()
    out = p.rollout(5, 1, constant_action=1)
    s, d = p.batch_stats()
    rew = p.get_obs()[1]
    assert s == _stats_order(rew) and (s, d) != s0
    p.step(np.ones(n, np.int32), 1)
    assert p.batch_stats()[0] == _stats_order(p.get_obs()[1])
    p.close()


def test_a_captured_graph_stays_correct_after_the_host_state_it_was_recorded_under_has_changed():
    """host-side decisions evaluated at enqueue time (the bare levels' static_charge, the batch scalars' freshness)
    would be frozen into a captured launch.  A step of the BARE kernel and a batch-stats request are captured while every
    battery is charged, then bsk_set_state empties some batteries: the replayed graph must report them (battery-empty
    termination, obs[3] = 0) and the replayed stats request must describe the replayed step, not the one before."""
    import torch
    n = 1000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cfg = default_config(4, GRAV_PM_J2)
        p = BatchedPropagator(cfg, n, stream=side.cuda_stream)
        ic = sample_ic_batch(n, 4, seed=23)
        p.reset(ic)
        act = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(3):
            p.step_device(act.data_ptr(), 1)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            p.step_device(act.data_ptr(), 1)
            stats_ptr = p.batch_stats_device()
        graph.replay()
        torch.cuda.synchronize()
        obs, rew, done, why = p.get_obs()
        assert not (why & 4).any()
        st = p.get_state()
        t = 12 + 4
        st[t + 7, ::10] = 0.0                                       # every tenth battery is empty now
        p.set_state(st)
        graph.replay()
        torch.cuda.synchronize()
        obs, rew, done, why = p.get_obs()
        assert np.array_equal((why & 4) != 0, np.arange(n) % 10 == 0) and np.all(obs[3, ::10] == 0.0) and np.all(obs[3, 1::10] > 0.0)
        s, d = p.batch_stats()
        assert s == _stats_order(rew) and d == int((why != 0).sum()) == n // 10
        del graph
        p.close()


def test_in_launch_wave_sums_are_not_trusted_once_a_graph_steps_the_handle():
    """"The last launch wrote the per-wave sums" is host-side knowledge and a replayed graph steps without telling the host: a handle
    whose launches have been captured forms the batch scalars from the reward buffer again (two-level form), whatever
    bsk_set_step_stats says - an eager step with the sums followed by a replayed step without them must not join stale sums."""
    import torch
    n = 5000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        p = BatchedPropagator(default_config(4, GRAV_PM_J2), n, stream=side.cuda_stream)
        p.reset(sample_ic_batch(n, 4, seed=41))
        a0 = torch.zeros(n, dtype=torch.int32, device="cuda")
        a1 = torch.ones(n, dtype=torch.int32, device="cuda")
        p.step_device(a0.data_ptr(), 1)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            p.step_device(a1.data_ptr(), 1)                # captured WITHOUT the sums
        p.set_step_stats(True)
        p.step_device(a0.data_ptr(), 1)                    # eager, action 0: rewards > 0
        s0, _ = p.batch_stats()
        assert s0 == _stats_order(p.get_obs()[1]) and s0 > 0.0
        graph.replay()                                     # action 1: every reward is 0
        torch.cuda.synchronize()
        s1, _ = p.batch_stats()
        rew = p.get_obs()[1]
        assert np.all(rew == 0.0) and s1 == 0.0 == _stats_order(rew)
        del graph
        p.close()


def test_reset_then_stats_on_a_captured_handle_report_the_last_steps_sums():
    """on a handle whose launches live in a HIP graph the host's "snapshot is fresh" flag is not trusted, so a
    stats request used to re-form the sums from the reward buffer - AFTER a reset entry point had zeroed the restarted envs' rewards:
    their terminal rewards and penalties dropped out.  The reset now seals the snapshot on the device (bsk_aux.hip: stats_sealed:
    env 0's counter word + episode number, changed by every step launch) and the join kernel leaves a sealed snapshot alone:
    step (graph) -> reset_from_pool_device(mask) -> stats = the step's sums; a further replayed step lifts the seal."""
    import torch
    n = 3000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cfg = default_config(4, GRAV_PM_J2)
        cfg.max_length = 2
        cfg.flags |= FLAG_AUTO_RESET
        p = BatchedPropagator(cfg, n, stream=side.cuda_stream)
        p.reset(sample_ic_batch(n, 4, seed=51))
        p.set_ic_pool(sample_ic_batch(64, 4, seed=52))
        act = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(2):
            p.step_device(act.data_ptr(), 1)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            p.step_device(act.data_ptr(), 1)
        graph.replay()                                      # the third step: max_length reached, every env finishes
        torch.cuda.synchronize()
        obs, rew, done, why = p.get_obs()
        want_s, want_d = _stats_order(rew), int((why != 0).sum())
        assert want_d == n and want_s > 0.0
        mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
        mask[::2] = 1                                        # env 0 among the restarted ones
        p.reset_from_pool_device(mask.data_ptr())
        torch.cuda.synchronize()
        assert np.all(p.get_obs()[1][::2] == 0.0) and np.all(p.get_obs()[1][1::2] > 0.0)
        assert p.batch_stats() == (want_s, want_d)           # the snapshot, not the zeroed buffer's sums
        sp = p.batch_stats_device()                          # ... and through the device-resident entry
        torch.cuda.synchronize()
        assert p.batch_stats() == (want_s, want_d)
        mask[:] = 0
        mask[1::2] = 1                                       # env 0 NOT among them: the seal rests on its unchanged counters
        p.reset_from_pool_device(mask.data_ptr())
        torch.cuda.synchronize()
        assert np.all(p.get_obs()[1] == 0.0) and p.batch_stats() == (want_s, want_d)
        graph.replay()                                       # a step nobody told the host about lifts the seal
        torch.cuda.synchronize()
        obs, rew, done, why = p.get_obs()
        s2, d2 = p.batch_stats()
        assert s2 == _stats_order(rew) and d2 == int((why != 0).sum()) and (s2, d2) != (want_s, want_d)
        del graph, sp
        p.close()


def test_a_captured_stats_request_alone_never_freezes_the_join_only_form():
    """bsk_set_step_stats(1), an eager step (it writes the per-wave sums), then a capture that holds ONLY the stats
    request: the join-only form must not be recorded - replays follow steps that no longer write the sums (a replayable handle
    steps in the two-level form)."""
    import torch
    n = 4096 + 17
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        p = BatchedPropagator(default_config(4, GRAV_PM_J2), n, stream=side.cuda_stream)
        p.reset(sample_ic_batch(n, 4, seed=61))
        a0 = torch.zeros(n, dtype=torch.int32, device="cuda")
        a1 = torch.ones(n, dtype=torch.int32, device="cuda")
        p.set_step_stats(True)
        p.step_device(a0.data_ptr(), 1)                     # eager, writes the wave sums
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            p.batch_stats_device()                           # the request alone
        graph.replay()
        torch.cuda.synchronize()
        s0, _ = p.batch_stats()
        assert s0 == _stats_order(p.get_obs()[1]) and s0 > 0.0
        p.step_device(a1.data_ptr(), 1)                     # replayable handle: no in-launch sums any more; rewards all 0
        graph.replay()
        torch.cuda.synchronize()
        dptr = p.batch_stats_device()
        torch.cuda.synchronize()
        rew = p.get_obs()[1]
        assert np.all(rew == 0.0) and p.batch_stats()[0] == 0.0
        del graph, dptr
        p.close()


def test_step_tensors_loop_is_hip_graph_capturable():
    """The device-resident loop - policy kernels + step kernel + device-side auto-reset - captured in a HIP graph and replayed
    gives exactly what the eager loop gives: step_tensors launches on the capturing stream and issues nothing a capture
    forbids (no copy, no synchronisation, no allocation)."""
    import torch
    n, U, reps = 1000, 5, 8
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        def make():
            probe = LeoPowerAttVecEnv(n, n_rw=4, step_duration=1.0, seed=3, device_reset_pool=128, device_sampler=True)
            cfg = probe.cfg
            probe.close()
            cfg.max_length = 7                 # episodes end - and restart on the device - inside the replayed graphs
            env = LeoPowerAttVecEnv(n, cfg=cfg, step_duration=1.0, seed=3, device_reset_pool=128, device_sampler=True, stream=side.cuda_stream)
            env.reset_tensors()
            return env
        g = torch.Generator(device="cuda").manual_seed(0)
        w = torch.randn(5, 3, dtype=torch.float64, device="cuda", generator=g)
        act = torch.zeros(n, dtype=torch.int64, device="cuda")
        logits = torch.zeros(n, 3, dtype=torch.float64, device="cuda")

        def one(env):
            torch.matmul(env._torch_views()["obs_n51"].reshape(n, 5), w, out=logits)
            torch.argmax(logits, dim=1, out=act)
            return env.step_tensors(act)

        env = make()
        for _ in range(U * (reps + 1)):
            ob, rew, done, info = one(env)
        torch.cuda.synchronize()
        want = [t.clone() for t in (ob, rew, info["episode_return"], info["episodes"])]
        env.close()

        env = make()
        for _ in range(U):                     # warm-up: torch's lazy initialisation must not happen inside the capture
            one(env)
        torch.cuda.synchronize()
        c0 = env.propagator.debug_counters()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(U):
                ob, rew, done, info = one(env)
        for _ in range(reps - 1):              # (the capture itself executed nothing)
            graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        assert env.propagator.debug_counters() == c0
        got = [ob, rew, info["episode_return"], info["episodes"]]
        assert int(info["episodes"].sum()) > 0
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        del graph
        env.close()


def test_device_views_dlpack_and_cuda_array_interface_alias_the_buffers():
    import torch
    n = 200
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= FLAG_AUTO_RESET
    p = BatchedPropagator(cfg, n)
    p.set_ic_pool(sample_ic_batch(8, 4, seed=0))
    p.reset(sample_ic_batch(n, 4, seed=1))
    p.step(np.zeros(n, np.int32), 3)
    obs, rew, done, why = p.get_obs()
    v = p.device_views()
    assert set(v) >= {"obs", "reward", "reason", "done_mask", "state", "terminal_obs", "episodes", "stride"}
    for name, want in (("obs", obs), ("reward", rew), ("reason", why), ("state", p.get_state())):
        a = torch.from_dlpack(v[name])
        b = torch.as_tensor(v[name], device="cuda")
        assert a.is_cuda and a.data_ptr() == b.data_ptr() and a.stride() == b.stride()
        assert np.array_equal(a.cpu().numpy(), want)
    assert v["obs"].__dlpack_device__() == (10, 0)
    assert torch.from_dlpack(v["episodes"]).dtype == torch.int32
    assert p.stream_ptr() != 0
    p.close()


@pytest.mark.parametrize("n", [130, 1000])
def test_sharded_vec_env_two_handles_one_card_equals_unsharded(n):
    kw = dict(n_rw=3, gravity_model=GRAV_PM_J2, step_duration=2.0, seed=11, device_reset_pool=32)
    probe = LeoPowerAttVecEnv(n, **kw)
    cfg = probe.cfg
    probe.close()
    cfg.max_length = 3
    one = LeoPowerAttVecEnv(n, cfg=cfg, step_duration=2.0, seed=11, device_reset_pool=32)
    two = ShardedVecEnv(n, devices=[0, 0], cfg=cfg, step_duration=2.0, seed=11, device_reset_pool=32)
    assert len(two.propagator.shards) == 2 and two.propagator.shards[0].stream_ptr() != two.propagator.shards[1].stream_ptr()
    assert np.array_equal(one.reset(), two.reset())
    rng = np.random.default_rng(5)
    for _ in range(7):
        a = rng.integers(0, 3, n)
        r1, r2 = one.step(a), two.step(a)
        for x, y in zip(r1[:3], r2[:3]):
            assert np.array_equal(x, y)
        for i in np.flatnonzero(r1[2]):
            assert np.array_equal(r1[3][i]["terminal_observation"], r2[3][i]["terminal_observation"])
            assert r1[3][i]["episode"] == r2[3][i]["episode"]
    assert np.array_equal(one.get_state(), two.get_state())
    assert one.batch_stats()[1] == two.batch_stats()[1]
    one.close()
    two.close()


def test_sharded_gather_obs_device_single_shard_path():
    """world = 1: the direct gather degenerates to the root's own strided device-to-device copy (the RCCL legs need
    distinct GPUs; their address arithmetic is covered on CPU, tests/test_sharded_host.py)."""
    import torch
    n = 300
    cfg = default_config(4, GRAV_PM_J2)
    sp = ShardedPropagator(cfg, n, devices=[0])
    sp.reset(sample_ic_batch(n, 4, seed=2))
    sp.step(np.zeros(n, np.int32), 5)
    view = sp.gather_obs_device(root=0)
    t = torch.from_dlpack(view)              # drains the handle's stream, then aliases the gather buffer
    obs = sp.get_obs()[0]
    assert tuple(t.shape) == (5, n) and t.is_contiguous() and np.array_equal(t.cpu().numpy(), obs)
    sp.close()


def test_get_obs_state_is_the_two_read_backs_in_one():
    n = 77
    cfg = default_config(3, GRAV_PM_J2)
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 3, seed=9))
    p.step((np.arange(n) % 3).astype(np.int32), 25)
    obs, st = p.get_obs_state()
    assert np.array_equal(obs, p.get_obs()[0]) and np.array_equal(st, p.get_state())
    p.close()


@pytest.mark.parametrize("n", [500, 512, 65536])
def test_read_back_forms_agree(n):
    """bsk_get_obs into fresh arrays, into the propagator's page-locked block (one contiguous copy when the batch fills its rows:
    n a multiple of 256), bsk_get_obs_rowmajor (the kernel's own (N, 5) block) and bsk_get_obs_state: the same numbers."""
    from basilisk_env_amd._lib import FLAG_OBS_ROWMAJOR
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= FLAG_OBS_ROWMAJOR
    cfg.max_length = 2
    p = BatchedPropagator(cfg, n)
    p.reset(sample_ic_batch(n, 4, seed=51))
    rng = np.random.default_rng(n)
    for _ in range(3):
        p.step(rng.integers(0, 3, n).astype(np.int32), 2)
        obs, rew, done, why = p.get_obs()
        o2, r2, d2, w2 = p.get_obs(copy=False)
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(done, d2) and np.array_equal(why, w2)
        o3, r3, d3, w3 = p.get_obs_rowmajor()
        assert np.array_equal(obs.T, o3) and np.array_equal(rew, r3) and np.array_equal(done, d3) and np.array_equal(why, w3)
        o4, st = p.get_obs_state()
        assert np.array_equal(obs, o4) and np.array_equal(st, p.get_state())
    assert done.any()
    q = BatchedPropagator(default_config(4, GRAV_PM_J2), 64)
    assert q.get_obs_rowmajor() is None                     # (no row-major block without the flag)
    p.close(); q.close()


def test_fp64_calibration_reports_a_plausible_sustained_rate():
    """bsk_calibrate_fp64: independent v_fma_f64 chains; two waves per SIMD issue faster than one, and neither exceeds
    the nominal 78.6 TFLOP/s of the part."""
    from basilisk_env_amd._lib import BskError, calibrate_fp64
    tf1, ns1 = calibrate_fp64(0, 1, 3)
    tf2, ns2 = calibrate_fp64(0, 2, 3)
    assert 20.0 < tf1 < 79.0 and 20.0 < tf2 < 79.0 and tf2 >= 0.95 * tf1
    assert 1.0 < ns2 <= ns1 * 1.05 < 5.0
    with pytest.raises(BskError):
        calibrate_fp64(0, 0, 3)
