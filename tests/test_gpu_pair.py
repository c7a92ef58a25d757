"""GPU: the pair form of the step kernel (two cooperating waves per 64 spacecraft: a dynamics wave and an FSW + environment
wave exchanging through LDS; bsk_device.hpp: PairLds) against the single-wave form on the same inputs.  Same arithmetic per
spacecraft in the same chunks of the tick loop (the Sun's tidal carry is anchored per chunk), so the results must be IDENTICAL
bit for bit at every level.  Both forms are held to the oracle and the 50-digit goldens by the rest of the suite (the pair form is what small batches run by default for launches of >= 16 sub-steps)."""
import os

import numpy as np
import pytest

from basilisk_env_amd._lib import FLAG_AUTO_RESET, FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY, GRAV_PM, GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch

pytestmark = pytest.mark.gpu


def make(cfg, n, pair):
    """A propagator pinned to the pair form (every launch) or to the single-wave form; the three-wave form
    (tests/test_gpu_tri.py), which the full-scenario level would otherwise prefer, is off."""
    old = {k: os.environ.get(k) for k in ("BSKGPU_PAIR", "BSKGPU_TRI")}
    os.environ["BSKGPU_PAIR"] = "1" if pair else "0"
    os.environ["BSKGPU_TRI"] = "0"
    try:
        return BatchedPropagator(cfg, n)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


LEVELS = {"power": FLAG_POWER, "full-nosun": FLAG_POWER | FLAG_DRAG | FLAG_DESAT,
          "full": FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT}


@pytest.mark.parametrize("level", ["power", "full-nosun", "full"])
@pytest.mark.parametrize("n_rw,grav", [(4, GRAV_PM_J2), (3, GRAV_PM), (0, GRAV_PM_J2)])
@pytest.mark.parametrize("lags", [(1, 1), (0, 0), (1, 0)])
def test_pair_form_equals_single_wave_form(level, n_rw, grav, lags):
    if n_rw == 0 and level != "power":
        pytest.skip("desaturation needs wheels")
    n = 333
    cfg = default_config(n_rw, grav)
    cfg.flags |= LEVELS[level] & (~FLAG_DESAT if n_rw == 0 else ~0)
    cfg.fsw_lag, cfg.nav_lag = lags
    if level != "power":
        cfg.base_density, cfg.scale_height = 1e-9, 100e3        # drag that matters at 500 km
    ic = sample_ic_batch(n, n_rw, seed=17)
    if n_rw:
        ic[12:12 + n_rw, ::5] *= 4.0                             # some wheels above the dumping threshold
    a, b = make(cfg, n, False), make(cfg, n, True)
    a.reset(ic)
    b.reset(ic)
    rng = np.random.default_rng(4)
    tol = 0.0
    for call, k in enumerate((1, 16, 20, 37, 3, 180, 7, 64)):     # below and above one FSW period, not multiples of the chunk
        act = rng.integers(0, 3, n).astype(np.int32)
        a.step(act, k)
        b.step(act, k)
        if call == 3:                                             # stagger the FSW phases inside the waves
            mask = (rng.random(n) < 0.3).astype(np.uint8)
            fresh = sample_ic_batch(n, n_rw, seed=99)
            a.reset(fresh, mask)
            b.reset(fresh, mask)
        sa, sb = a.get_state(), b.get_state()
        scale = np.maximum(np.abs(sa).max(axis=1, keepdims=True), 1e-300)
        assert (np.abs(sa - sb) / scale).max() <= tol, (level, n_rw, lags, call, k)
        for x, y in zip(a.get_obs(), b.get_obs()):
            assert np.array_equal(x, y) if tol == 0.0 else np.abs(x.astype(float) - y.astype(float)).max() <= 1e-12
        assert all(np.array_equal(x, y) for x, y in zip(a.get_counters(), b.get_counters()))
        assert a.batch_stats()[1] == b.batch_stats()[1]
    assert "pair" in b.kernel_info()["name"] and "pair" not in a.kernel_info()["name"]
    a.close()
    b.close()


def test_pair_form_with_device_side_reset_and_default_selection():
    """Auto-reset epilogue under the pair form; and the default rule: the pair form for launches of >= 16 sub-steps of small
    batches, the single-wave form otherwise."""
    n = 200
    cfg = default_config(4, GRAV_PM_J2)
    cfg.flags |= LEVELS["full-nosun"] | FLAG_AUTO_RESET
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
    pcfg = default_config(4, GRAV_PM_J2)              # default rule at the power level (the full-scenario level prefers the
    pcfg.flags |= FLAG_POWER                          # three-wave form: tests/test_gpu_tri.py)
    p = BatchedPropagator(pcfg, n)
    p.reset(ic)
    p.step(act, 20)
    assert "pair" in p.kernel_info()["name"]
    p.step(act, 3)
    assert "pair" not in p.kernel_info()["name"]
    p.close()
    big = BatchedPropagator(default_config(4, GRAV_PM_J2), 64)     # bare level: no pair form
    big.reset(sample_ic_batch(64, 4, seed=1))
    big.step(np.zeros(64, np.int32), 40)
    assert "pair" not in big.kernel_info()["name"]
    big.close()
