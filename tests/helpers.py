"""Shared helpers for the parity tests."""
import numpy as np


def field_groups(n_rw):
    g = {"r": slice(0, 3), "v": slice(3, 6), "sigma": slice(6, 9), "omega": slice(9, 12)}
    if n_rw:
        g["Omega"] = slice(12, 12 + n_rw)
        g["u"] = slice(12 + n_rw + 3, 12 + n_rw + 3 + n_rw)
        g["u_pend"] = slice(12 + n_rw + 26, 12 + n_rw + 26 + n_rw)
        g["sigma_BR_msg"] = slice(12 + n_rw + 30, 12 + n_rw + 31)
    return g


def rel_err(a, b, sl):
    """max |a-b| over the group, relative to the group's largest magnitude in b."""
    return float(np.abs(a[sl] - b[sl]).max() / max(np.abs(b[sl]).max(), 1e-300))


def max_group_err(a, b, n_rw):
    return {k: rel_err(a, b, sl) for k, sl in field_groups(n_rw).items()}


def cfg_for_case(case):
    """bsk_config of a golden case (tests/golden/make_golden.py: the recipes' cfg_edit functions)."""
    from basilisk_env_amd._lib import FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
    from basilisk_env_amd.simulators.dynamics.config import default_config
    cfg = default_config(case["n_rw"], case["gravity_model"])
    if case.get("cfg_edit") == "scenario_edit":
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        cfg.base_density, cfg.scale_height = 1e-9, 100e3
    if case.get("cfg_edit") == "dense_drag_edit":
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        cfg.dt = 1.0
        cfg.base_density, cfg.scale_height = 1e-4, 20e3
    if case.get("cfg_edit") == "desat_edit":
        from basilisk_env_amd._lib import FLAG_DESAT
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT
    if case.get("cfg_edit") == "nolag_edit":
        cfg.fsw_lag = 0
        cfg.nav_lag = 0
    if case.get("cfg_edit") == "navnow_edit":
        cfg.nav_lag = 0
    if "sh_degree" in case:
        cfg.sh_degree = case["sh_degree"]
    return cfg
