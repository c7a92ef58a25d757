"""ctypes binding of libbskgpu.so — the C-ABI declared in include/bskgpu.h.

The library is the only compute path: if it is missing, fails to load, or no gfx950 device is
visible, the product raises (:class:`BskGpuUnavailable`); there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

BSK_ABI_VERSION = 4
BSK_MAX_RW = 4
BSK_MAX_THR = 8

GRAV_PM, GRAV_PM_J2, GRAV_SH = 0, 1, 2
FLAG_SUN_THIRD_BODY, FLAG_POWER, FLAG_DESAT, FLAG_DRAG, FLAG_AUTO_RESET, FLAG_LDS_SCRATCH = 1, 2, 4, 8, 16, 32
FLAG_EPISODE_STATS, FLAG_OBS_ROWMAJOR = 64, 128
DONE_LENGTH, DONE_WHEELS, DONE_BATTERY, DONE_ORBIT = 1, 2, 4, 8

# state field offsets (include/bskgpu.h)
F_R, F_V, F_SIGMA, F_OMEGA, NF_BASE = 0, 3, 6, 9, 12
T_LEXT, T_UCMD, T_CHARGE, T_THR_REM, T_THR_LIM, T_THR_T0, T_THR_CNT, T_UPEND, T_SBR, NF_TAIL = 0, 3, 7, 8, 16, 24, 25, 26, 30, 31


def n_fields(n_rw):
    return NF_BASE + n_rw + NF_TAIL


class BskGpuUnavailable(RuntimeError):
    """libbskgpu.so cannot be used (not built, not loadable, or no gfx950 device)."""


class BskError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbskgpu error %d: %s" % (code, msg))
        self.code = code


d, i32, u32 = C.c_double, C.c_int32, C.c_uint32


class BskConfig(C.Structure):
    """Mirror of ``struct bsk_config`` (include/bskgpu.h) — field order and types must match."""
    _fields_ = [
        ("abi_version", u32), ("struct_size", u32),
        ("dt", d), ("fsw_every", i32), ("gravity_model", i32), ("sh_degree", i32), ("n_rw", i32),
        ("flags", u32), ("max_length", i32), ("fsw_lag", i32), ("nav_lag", i32),
        ("mu", d), ("req", d), ("j2", d), ("planet_rate", d),
        ("inertia", d * 9), ("mass", d),
        ("gs", (d * 3) * BSK_MAX_RW), ("js", d * BSK_MAX_RW), ("u_max", d), ("u_min", d), ("f_coulomb", d),
        ("K", d), ("P", d), ("sigma_R0N", d * 3), ("ctrl_axes", d * 9),
        ("wheel_limit", d), ("power_max", d), ("reward_mult", d), ("failure_penalty", d), ("r_min", d),
        ("panel_normal", d * 3), ("panel_area", d), ("panel_efficiency", d), ("power_draw", d),
        ("storage_capacity", d), ("solar_flux", d),
        ("sun_r0", d * 3), ("sun_v", d * 3), ("mu_sun", d),
        ("n_thr", i32), ("thr_max_counter", i32), ("thr_pos", (d * 3) * BSK_MAX_THR), ("thr_dir", (d * 3) * BSK_MAX_THR),
        ("thr_max_thrust", d), ("thr_min_fire_time", d), ("thr_min_on_time", d), ("hs_min", d),
        ("base_density", d), ("scale_height", d), ("n_facets", i32), ("pad0_", i32),
        ("facet_area", d * 8), ("facet_cd", d * 8), ("facet_normal", (d * 3) * 8), ("facet_pos", (d * 3) * 8),
    ]

    def copy(self):
        out = BskConfig()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(BskConfig))
        return out


EXPORTS = [
    "bsk_default_config", "bsk_create", "bsk_destroy", "bsk_set_gravity_sh", "bsk_reset", "bsk_step",
    "bsk_step_device", "bsk_step_device_i64", "bsk_step_n", "bsk_get_episode_device", "bsk_get_batch_stats_device", "bsk_set_step_stats", "bsk_reset_from_pool_device", "bsk_debug_counters", "bsk_debug_words", "bsk_get_obs", "bsk_get_obs_rowmajor", "bsk_get_obs_device", "bsk_get_obs_state", "bsk_get_stream", "bsk_get_terminal_obs_device", "bsk_get_state_device", "bsk_get_batch_stats", "bsk_n_fields",
    "bsk_get_state", "bsk_set_state", "bsk_get_counters", "bsk_set_counters", "bsk_set_ic_pool", "bsk_sample_ic_pool", "bsk_reset_from_pool", "bsk_get_ic_pool", "bsk_get_terminal_obs", "bsk_set_env_base", "bsk_set_sim_time", "bsk_sync",
    "bsk_profile_begin", "bsk_profile_set_stride", "bsk_profile_end", "bsk_profile_end_samples", "bsk_calibrate_fp64", "bsk_kernel_info", "bsk_last_error", "bsk_version",
]

_LIB = None


def lib_path():
    """In-tree library; ``BSKGPU_LIB`` overrides it (kernel A/B experiments only)."""
    return os.environ.get("BSKGPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbskgpu.so")


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so``
    (SONAME libamdhip64.so.7, the SONAME libbskgpu.so asks for).  If libbskgpu.so were loaded
    first it would bind /opt/rocm's copy and a later ``import torch`` would bring a second runtime
    into the process: torch then sees no GPU and device pointers cannot be shared with RCCL.
    Preloading torch's copy (without importing torch) makes every import order end with one
    runtime; without torch installed the system ROCm runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return cand
    except Exception:
        return None
    return None


def load():
    """Load libbskgpu.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise BskGpuUnavailable(
            "%s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "basilisk_env_amd/csrc`). There is no CPU fallback for the propagator." % path)
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the host
        raise BskGpuUnavailable("cannot load %s: %s" % (path, e)) from e
    P = C.POINTER
    vp = C.c_void_p
    lib.bsk_last_error.restype = C.c_char_p
    lib.bsk_version.restype = C.c_char_p
    lib.bsk_default_config.argtypes = [P(BskConfig), C.c_int, C.c_int]
    lib.bsk_create.argtypes = [P(BskConfig), C.c_int, C.c_int, vp, P(vp)]
    lib.bsk_destroy.argtypes = [vp]
    lib.bsk_destroy.restype = None
    lib.bsk_set_gravity_sh.argtypes = [vp, C.c_int, vp, vp]
    lib.bsk_reset.argtypes = [vp, vp, vp]
    lib.bsk_step.argtypes = [vp, vp, C.c_int]
    lib.bsk_step_device.argtypes = [vp, vp, C.c_int]
    for name, args in (("bsk_step_device_i64", [vp, vp, C.c_int]), ("bsk_get_episode_device", [vp, P(vp), P(vp), P(vp), P(vp), P(vp)]),
                       ("bsk_get_batch_stats_device", [vp, P(vp)]), ("bsk_set_step_stats", [vp, C.c_int]), ("bsk_get_obs_rowmajor", [vp, vp, vp, vp]), ("bsk_reset_from_pool_device", [vp, vp]),
                       ("bsk_debug_counters", [P(C.c_int64), P(C.c_int64)]), ("bsk_debug_words", [vp, vp]),
                       ("bsk_step_n", [vp, vp, C.c_int32, C.c_int, C.c_int, vp, vp, vp])):
        # (a BSKGPU_LIB variant built from an older tree - kernel A/B against a previous round - may predate these)
        if hasattr(lib, name) or not os.environ.get("BSKGPU_LIB"):
            getattr(lib, name).argtypes = args
    lib.bsk_get_obs.argtypes = [vp, vp, vp, vp, vp]
    lib.bsk_get_obs_device.argtypes = [vp, P(vp), P(vp), P(vp), P(vp), P(C.c_int64)]
    lib.bsk_get_obs_state.argtypes = [vp, vp, vp, vp, vp]
    lib.bsk_get_stream.argtypes = [vp, P(vp)]
    lib.bsk_get_terminal_obs_device.argtypes = [vp, P(vp), P(vp)]
    lib.bsk_get_state_device.argtypes = [vp, P(vp), P(C.c_int64)]
    lib.bsk_get_batch_stats.argtypes = [vp, P(C.c_double), P(C.c_int64)]
    lib.bsk_n_fields.argtypes = [vp]
    lib.bsk_get_state.argtypes = [vp, vp]
    lib.bsk_set_state.argtypes = [vp, vp]
    lib.bsk_get_counters.argtypes = [vp, vp, vp]
    lib.bsk_set_counters.argtypes = [vp, vp, vp]
    lib.bsk_set_ic_pool.argtypes = [vp, C.c_int, vp]
    lib.bsk_get_terminal_obs.argtypes = [vp, vp, vp]
    lib.bsk_sample_ic_pool.argtypes = [vp, C.c_int, C.c_uint64]
    lib.bsk_reset_from_pool.argtypes = [vp, vp]
    lib.bsk_get_ic_pool.argtypes = [vp, vp]
    lib.bsk_set_sim_time.argtypes = [vp, C.c_double]
    lib.bsk_set_env_base.argtypes = [vp, C.c_int64]
    lib.bsk_sync.argtypes = [vp]
    lib.bsk_profile_begin.argtypes = [vp, C.c_int]
    lib.bsk_profile_set_stride.argtypes = [vp, C.c_int]
    lib.bsk_profile_end.argtypes = [vp, P(C.c_double), P(C.c_int)]
    lib.bsk_profile_end_samples.argtypes = [vp, P(C.c_double), P(C.c_int), vp, C.c_int]
    lib.bsk_calibrate_fp64.argtypes = [C.c_int, C.c_int, C.c_int, P(C.c_double), P(C.c_double)]
    lib.bsk_kernel_info.argtypes = [vp, C.c_char_p, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]
    _LIB = lib
    return lib


def calibrate_fp64(device=0, waves_per_simd=2, repeats=5):
    """-> (TFLOP/s, ns per FMA wave-instruction and SIMD) this device sustains on independent fp64 FMA chains."""
    tf, ns = C.c_double(), C.c_double()
    check(load().bsk_calibrate_fp64(int(device), int(waves_per_simd), int(repeats), C.byref(tf), C.byref(ns)))
    return tf.value, ns.value


def check(rc):
    if rc != 0:
        msg = load().bsk_last_error()
        msg = msg.decode("utf-8", "replace") if msg else ""
        if rc == -2:
            raise BskGpuUnavailable("libbskgpu: %s" % msg)
        raise BskError(rc, msg)
