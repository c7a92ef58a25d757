"""GPU: the three-wave form of the full-scenario step kernel (a translational wave, a rotational wave and the FSW + environment
wave per 64 spacecraft; the two halves of the integration exchange three doubles each way per RK4 stage through tagged LDS
slots, without a barrier: bsk_device.hpp: TriX, rk4_step_part) against the single-wave form on the same inputs.  Every value
either half produces is computed by the operations of the whole, in its order, so the results must be IDENTICAL bit for bit -
including where ticks with drag (the halves exchange every stage) alternate with ticks above the atmosphere (they exchange
once per tick) and thruster bursts.  The form is what batches of <= 16 384 spacecraft run for launches of >= 16 sub-steps."""
import os

import numpy as np
import pytest

from basilisk_env_amd._lib import FLAG_AUTO_RESET, FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

pytestmark = pytest.mark.gpu


def make(cfg, n, tri):
    old = {k: os.environ.get(k) for k in ("BSKGPU_PAIR", "BSKGPU_TRI")}
    os.environ["BSKGPU_TRI"] = "1" if tri else "0"
    os.environ["BSKGPU_PAIR"] = "0"
    try:
        return BatchedPropagator(cfg, n)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


FULL = FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT


def _same(a, b, ctx):
    sa, sb = a.get_state(), b.get_state()
    assert np.isfinite(sa).all(), ctx
    assert np.array_equal(sa, sb), (ctx, np.argwhere(sa != sb)[:5].tolist())
    for x, y in zip(a.get_obs(), b.get_obs()):
        assert np.array_equal(x, y), ctx
    assert all(np.array_equal(x, y) for x, y in zip(a.get_counters(), b.get_counters())), ctx
    assert a.batch_stats()[1] == b.batch_stats()[1], ctx


@pytest.mark.parametrize("atmosphere", ["thick", "reference", "none"])
@pytest.mark.parametrize("n_rw,grav", [(4, GRAV_PM_J2), (3, GRAV_PM)])
@pytest.mark.parametrize("lags", [(1, 1), (0, 0), (1, 0)])
def test_three_wave_form_equals_single_wave_form(atmosphere, n_rw, grav, lags):
    """thick: drag at every altitude (every tick couples the halves); reference: the reference's atmosphere, in which the waves
    of this batch cross the skip density back and forth (coupled and uncoupled ticks alternate inside a launch); none: the drag
    flag off (only the tick-boundary exchange ever runs).  Wheels above the dumping threshold fire thruster bursts."""
    n = 333
    cfg = default_config(n_rw, grav)
    cfg.flags |= FULL if atmosphere != "none" else FULL & ~FLAG_DRAG
    cfg.fsw_lag, cfg.nav_lag = lags
    if atmosphere == "thick":
        cfg.base_density, cfg.scale_height = 1e-9, 100e3
    ic = sample_ic_batch(n, n_rw, seed=17)
    ic[12:12 + n_rw, ::5] *= 4.0
    if atmosphere == "reference":
        # circular orbits a few km either side of the altitude where the reference's atmosphere reaches the skip density
        # (462 km), sorted so that some waves are wholly above, some wholly below and some straddle it
        alt = np.sort(np.random.default_rng(5).uniform(440e3, 480e3, n))
        r = cfg.req + alt
        ic[0], ic[1], ic[2] = r, 0.0, 0.0
        vc = np.sqrt(cfg.mu / r)
        ic[3], ic[4], ic[5] = -100.0, vc * np.cos(0.9), vc * np.sin(0.9)    # descending 100 m/s: the wave just above the
        #                                                                       threshold crosses it during the run
    a, b = make(cfg, n, False), make(cfg, n, True)
    a.reset(ic)
    b.reset(ic)
    rng = np.random.default_rng(4)
    for call, k in enumerate((1, 16, 20, 37, 3, 180, 7, 64)):
        act = rng.integers(0, 3, n).astype(np.int32)
        a.step(act, k)
        b.step(act, k)
        if call == 3:                                             # stagger the FSW phases inside the waves
            mask = (rng.random(n) < 0.3).astype(np.uint8)
            fresh = sample_ic_batch(n, n_rw, seed=99)
            a.reset(fresh, mask)
            b.reset(fresh, mask)
        _same(a, b, (atmosphere, n_rw, lags, call, k))
    assert "tri" in b.kernel_info()["name"] and "tri" not in a.kernel_info()["name"]
    a.close()
    b.close()


def test_three_wave_form_full_env_step_and_one_spacecraft():
    """The drop-in env's launch (1 800 sub-steps) for ONE spacecraft - the reference's own use - and for 64."""
    for n in (1, 64):
        cfg = default_config(3, GRAV_PM_J2)
        cfg.flags |= FULL
        ic = sample_ic_batch(n, 3, seed=21)
        a, b = make(cfg, n, False), make(cfg, n, True)
        a.reset(ic)
        b.reset(ic)
        for act in (0, 2, 1):
            a.step(np.full(n, act, np.int32), 1800)
            b.step(np.full(n, act, np.int32), 1800)
            _same(a, b, (n, act))
        a.close()
        b.close()


def test_three_wave_form_at_one_workgroup_per_cu():
    """16 384 spacecraft = 256 workgroups of 192 threads and 96 KB of LDS each, one per CU: the largest batch the form takes."""
    n = 16384
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= FULL
    ic = sample_ic_batch(n, 4, seed=8)
    ic[12:16, ::7] *= 4.0
    a, b = make(cfg, n, False), make(cfg, n, True)
    a.reset(ic)
    b.reset(ic)
    rng = np.random.default_rng(9)
    for k in (40, 25):
        act = rng.integers(0, 3, n).astype(np.int32)
        a.step(act, k)
        b.step(act, k)
        _same(a, b, (n, k))
    info = b.kernel_info()
    assert "tri" in info["name"] and info["grid"] == 256 and info["block"] == 192
    a.close()
    b.close()


def test_three_wave_form_with_device_side_reset_and_default_selection():
    n = 200
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= FULL | FLAG_AUTO_RESET
    cfg.max_length = 2
    pool = sample_ic_batch(16, 4, seed=3)
    ic = sample_ic_batch(n, 4, seed=2)
    a, b = make(cfg, n, False), make(cfg, n, True)
    for p in (a, b):
        p.set_ic_pool(pool)
        p.reset(ic)
    act = np.zeros(n, np.int32)
    for k in (20, 20, 20, 20):
        a.step(act, k)
        b.step(act, k)
        assert np.array_equal(a.get_state(), b.get_state())
        for x, y in zip(a.get_terminal_obs(), b.get_terminal_obs()):
            assert np.array_equal(x, y)
    a.close()
    b.close()
    p = BatchedPropagator(cfg, n)                     # default rule: launches of >= 16 sub-steps of small batches
    p.set_ic_pool(pool)
    p.reset(ic)
    p.step(act, 20)
    assert "tri" in p.kernel_info()["name"] and p.kernel_info()["block"] == 192
    p.step(act, 3)
    assert "tri" not in p.kernel_info()["name"] and "pair" not in p.kernel_info()["name"]
    p.close()
    big = BatchedPropagator(cfg, 16384 + 64)          # more than one workgroup per CU: not the three-wave form
    big.set_ic_pool(pool)
    big.reset(sample_ic_batch(16384 + 64, 4, seed=1))
    big.step(np.zeros(16384 + 64, np.int32), 20)
    assert "tri" not in big.kernel_info()["name"]
    big.close()


def test_exchange_timeout_raises_the_handles_error_word():
    """The barrier-free exchange gives up after 2^20 polls instead of hanging - and must SAY so: a probe library whose
    translational wave never publishes (csrc/bsk_probes.hpp: BSK_PROBE_TRI_NOPUBLISH, built by `make probes` /
    __graft_entry__.build()) makes the next synchronising call fail with BSK_EHIP and a message, not hand NaN observations
    to the caller as if nothing had happened.  Run in a child process: the library of this process is the product's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "basilisk_env_amd", "variants", "probe_tri_nopublish.so")
    if not os.path.exists(lib):
        pytest.skip("probe library not built (make -C basilisk_env_amd/csrc probes)")
    script = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from basilisk_env_amd._lib import BskError, FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
cfg = default_config(4, GRAV_PM_J2)
cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
p = BatchedPropagator(cfg, 64)
p.reset(sample_ic_batch(64, 4, seed=1))
p.step(np.zeros(64, np.int32), 20)
assert "tri" in p.kernel_info()["name"]
try:
    p.get_obs()
    print("NO ERROR")
except BskError as e:
    print("CODE", e.code, "three-wave" in str(e))
p.sync()                       # the word is cleared once reported
print("SYNC OK")
try:
    p.step(np.zeros(64, np.int32), 20)
    p.sync()
    print("NO ERROR")
except BskError as e:
    print("CODE", e.code)
p.close()
''' % root
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, BSKGPU_LIB=lib, BSKGPU_TRI="1", BSKGPU_PAIR="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.split("\n")
    assert out[0] == "CODE -4 True" and out[1] == "SYNC OK" and out[2] == "CODE -4", r.stdout
