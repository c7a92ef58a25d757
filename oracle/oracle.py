"""ctypes loader for the CPU oracle (oracle/bsk_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg —
never from the product package ``basilisk_env_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from basilisk_env_amd._lib import BskConfig, n_fields  # the ABI's config struct (data, not algorithm)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(_HERE, name)
    override = os.environ.get("BSK_ORACLE_LIB")          # tests/test_oracle_sanitizers.py: an instrumented build of the same source
    if override and not omp:
        path = override
    elif not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    P, vp = C.POINTER, C.c_void_p
    lib.orc_n_fields.argtypes = [P(BskConfig)]
    lib.orc_step.argtypes = [P(BskConfig), C.c_int, vp, vp, vp, vp, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp]
    lib.orc_gravity.argtypes = [P(BskConfig), vp, vp, vp, C.c_double, vp]
    lib.orc_eom.argtypes = [P(BskConfig), vp, vp, vp, C.c_double, vp]
    lib.orc_fsw.argtypes = [P(BskConfig), vp, C.c_int, vp, vp]
    lib.orc_mrp2c.argtypes = [vp, vp]
    lib.orc_c2mrp.argtypes = [vp, vp]
    lib.orc_submrp.argtypes = [vp, vp, vp]
    lib.orc_shadow.argtypes = [P(BskConfig), vp, vp]
    lib.orc_shadow.restype = C.c_double
    _LIBS[name] = lib
    return lib


def usable_cpus():
    """CPUs this process may actually use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def set_threads(n):
    lib = load(omp=True)
    lib.orc_set_threads(int(n))
    return lib.orc_get_threads()


def _p(a):
    return a.ctypes.data if a is not None else None


def step(cfg, state, steps, ticks, actions, substeps, sim_time0=0.0, cbar=None, sbar=None, omp=False):
    """One env step for every column of ``state`` ([n_fields, N], modified in place together with
    ``steps``/``ticks`` int32[N]).  -> obs (5,N), reward (N,), done (N,) bool, reason (N,) uint8."""
    lib = load(omp)
    n = state.shape[1]
    assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape[0] == n_fields(cfg.n_rw)
    assert steps.dtype == np.int32 and ticks.dtype == np.int32
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    obs = np.empty((5, n))
    rew = np.empty(n)
    done = np.empty(n, dtype=np.uint8)
    why = np.empty(n, dtype=np.uint8)
    cb = np.ascontiguousarray(cbar, dtype=np.float64) if cbar is not None else None
    sb = np.ascontiguousarray(sbar, dtype=np.float64) if sbar is not None else None
    rc = lib.orc_step(C.byref(cfg), n, _p(state), _p(steps), _p(ticks), _p(actions), int(substeps), float(sim_time0),
                      _p(cb), _p(sb), _p(obs), _p(rew), _p(done), _p(why))
    if rc != 0:
        raise RuntimeError("oracle rejected the configuration")
    return obs, rew, done.astype(bool), why


def gravity(cfg, r, t=0.0, cbar=None, sbar=None):
    lib = load()
    r = np.ascontiguousarray(r, dtype=np.float64)
    a = np.empty(3)
    cb = np.ascontiguousarray(cbar, dtype=np.float64) if cbar is not None else None
    sb = np.ascontiguousarray(sbar, dtype=np.float64) if sbar is not None else None
    if lib.orc_gravity(C.byref(cfg), _p(cb), _p(sb), _p(r), float(t), _p(a)) != 0:
        raise RuntimeError("oracle rejected the configuration")
    return a


def eom(cfg, x, u, lext, t=0.0):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64) if cfg.n_rw else np.zeros(1)
    lext = np.ascontiguousarray(lext, dtype=np.float64)
    dx = np.empty(12 + cfg.n_rw)
    if lib.orc_eom(C.byref(cfg), _p(x), _p(u), _p(lext), float(t), _p(dx)) != 0:
        raise RuntimeError("oracle rejected the configuration")
    return dx


def fsw(cfg, x, action):
    """-> dict(sigma_BR, omega_BR_B, omega_RN_B, domega_RN_B), u (n_rw,)"""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    g = np.empty(12)
    u = np.empty(max(cfg.n_rw, 1))
    if lib.orc_fsw(C.byref(cfg), _p(x), int(action), _p(g), _p(u)) != 0:
        raise RuntimeError("oracle rejected the configuration")
    return {"sigma_BR": g[0:3], "omega_BR_B": g[3:6], "omega_RN_B": g[6:9], "domega_RN_B": g[9:12]}, u[:cfg.n_rw]


def mrp2c(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    c = np.empty(9)
    load().orc_mrp2c(_p(q), _p(c))
    return c.reshape(3, 3)


def c2mrp(c):
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(9)
    q = np.empty(3)
    load().orc_c2mrp(_p(c), _p(q))
    return q


def submrp(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    q = np.empty(3)
    load().orc_submrp(_p(a), _p(b), _p(q))
    return q


def shadow(cfg, r, sun, omp=False):
    r = np.ascontiguousarray(r, dtype=np.float64)
    sun = np.ascontiguousarray(sun, dtype=np.float64)
    return load(omp).orc_shadow(C.byref(cfg), _p(r), _p(sun))


DECISIONS = {"friction_per_stage": 0, "sun_per_tick": 1, "t0_real_messages": 2}


def set_decision(name, value):
    """Flip one of the engine behaviours the restatement had to decide (all default 0 = what the
    kernels implement): 'friction_per_stage', 'sun_per_tick', 't0_real_messages'."""
    for omp in (False, True):
        if load(omp).orc_set_decision(DECISIONS[name], int(value)) != 0:
            raise ValueError(name)


def set_penumbra_form(form):
    """0 (default): the conditioned lens-area form the kernels are held to; 1: the expression exactly as the
    reference engine's eclipse module writes it, in fp64.  Applies to shadow() and step() of both
    oracle libraries."""
    for omp in (False, True):
        load(omp).orc_set_penumbra_form(int(form))
