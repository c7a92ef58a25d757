"""Scenario constants -> ``bsk_config``.

Every value is the one the reference scenario sets (file:line under /root/reference):
integrator/FSW rates ``envs/leoPowerAttitudeEnvironment.py:185``; hub mass and cuboid inertia
``simulators/leoPowerAttitudeSimulator.py:129,137-139,245-247``; gains ``:178-180``;
``sigma_R0N`` ``:170``; control axes ``:173-175``; power parameters ``:158-167``; env constants
``envs/leoPowerAttitudeEnvironment.py:25,36-42``; mu ``initial_conditions/leo_orbit.py:30``.
"""
import math

import numpy as np

from ..._lib import BSK_ABI_VERSION, GRAV_PM, BskConfig
from .effectorPrimatives import actuatorPrimatives as ap
import ctypes

MU_EARTH = 0.3986004415e15      # m^3/s^2, leo_orbit.py:12,30
REQ_EARTH_KM = 6378.1366        # Basilisk orbitalMotion.REQ_EARTH [km] (used at ...Simulator.py:146,641)
CBAR_20 = -4.841693e-4          # normalised degree-2 zonal used for J2 = -sqrt(5) * C20
RPM = 2.0 * math.pi / 60.0      # Basilisk macros.RPM
D2R = math.pi / 180.0
AU = 149597870700.0
EPOCH_JD = 2459338.5 + (7.0 * 3600.0 + 47.0 * 60.0 + 48.965) / 86400.0  # '2021 MAY 04 07:47:48.965 (UTC)', ...Simulator.py:219


FACETS = [
    (0.2 * 0.3, 2.2, [1, 0, 0], [0.05, 0.0, 0]),
    (0.2 * 0.3, 2.2, [-1, 0, 0], [0.05, 0.0, 0]),
    (0.1 * 0.2, 2.2, [0, 1, 0], [0, 0.15, 0]),
    (0.1 * 0.2, 2.2, [0, -1, 0], [0, -0.15, 0]),
    (0.1 * 0.3, 2.2, [0, 0, 1], [0, 0, 0.1]),
    (0.1 * 0.3, 2.2, [0, 0, -1], [0, 0, -0.1]),
    (1. * 2., 2.2, [0, 1, 0], [0, 2., 0]),
    (1. * 2., 2.2, [0, -1, 0], [0, 2., 0]),
]


def sun_position(jd):
    """Low-precision solar position (Astronomical Almanac), equatorial frame, metres, Earth-centred.
    Stands in for the SPICE de430 lookup of the reference (...Simulator.py:219-225)."""
    n = jd - 2451545.0
    L = math.fmod(280.460 + 0.9856474 * n, 360.0)
    g = math.fmod(357.528 + 0.9856003 * n, 360.0) * D2R
    lam = (L + 1.915 * math.sin(g) + 0.020 * math.sin(2 * g)) * D2R
    eps = (23.439 - 0.0000004 * n) * D2R
    R = (1.00014 - 0.01671 * math.cos(g) - 0.00014 * math.cos(2 * g)) * AU
    return np.array([R * math.cos(lam), R * math.cos(eps) * math.sin(lam), R * math.sin(eps) * math.sin(lam)])


def default_config(n_rw=3, gravity_model=GRAV_PM, mass=330.0, width=1.38, depth=1.04, height=1.58):
    """The reference scenario's constants for ``n_rw`` wheels (3: triad, 4: pyramid, 0: none)."""
    if n_rw not in (0, 3, 4):
        raise ValueError("n_rw must be 0, 3 or 4")
    c = BskConfig()
    c.abi_version = BSK_ABI_VERSION
    c.struct_size = ctypes.sizeof(BskConfig)
    c.dt = 0.1
    c.fsw_every = 10
    c.gravity_model = gravity_model
    c.n_rw = n_rw
    c.max_length = 540
    c.fsw_lag = 1   # mrpControlTask order of the reference (...Simulator.py:484-486): control lags guidance by one FSW tick
    c.nav_lag = 1   # FSW task priorities 100 / 50 against the dynamics tasks' default (:383-386, :101-103): FSW runs first
    c.mu = MU_EARTH
    c.req = REQ_EARTH_KM * 1000.0
    c.j2 = math.sqrt(5.0) * -CBAR_20
    c.planet_rate = 7.2921159e-5
    c.mass = mass
    c.inertia[0] = 1. / 12. * mass * (width * width + depth * depth)
    c.inertia[4] = 1. / 12. * mass * (depth * depth + height * height)
    c.inertia[8] = 1. / 12. * mass * (width * width + height * height)
    wheels = {0: [], 3: ap.balancedHR16Triad(), 4: ap.balancedHR16Pyramid()}[n_rw]
    for i, w in enumerate(wheels):
        for k in range(3):
            c.gs[i][k] = w.gsHat_B[k]
        c.js[i] = w.Js
    ref = ap.HONEYWELL_HR16
    c.u_max, c.u_min, c.f_coulomb = ref["u_max"], ref["u_min"], ref["fCoulomb"]
    c.K, c.P = 7.0, 35.0
    c.sigma_R0N[0] = 1.0
    c.ctrl_axes[0] = c.ctrl_axes[4] = c.ctrl_axes[8] = 1.0
    c.wheel_limit = 3000.0 * RPM
    c.power_max = 20.0
    c.reward_mult = 1.0 / 540.0
    c.failure_penalty = 1.0
    c.r_min = REQ_EARTH_KM / 1000.0
    c.panel_normal[1] = -1.0
    c.panel_area = 0.2 * 0.3
    c.panel_efficiency = 0.20
    c.power_draw = -5.0
    c.storage_capacity = 20.0 * 3600.0
    c.solar_flux = 1372.5398
    p0, p1 = sun_position(EPOCH_JD), sun_position(EPOCH_JD + 1.0)
    for k in range(3):
        c.sun_r0[k] = p0[k]
        c.sun_v[k] = (p1[k] - p0[k]) / 86400.0
    c.mu_sun = 1.32712440018e20
    c.hs_min = 4.0
    c.thr_max_counter = 4
    c.thr_min_fire_time = 0.002
    thr = ap.idealMonarc1Octet()
    c.n_thr = len(thr)
    for i, t in enumerate(thr):
        for k in range(3):
            c.thr_pos[i][k] = t.pos_B[k]
            c.thr_dir[i][k] = t.dir_B[k]
    c.thr_max_thrust = thr[0].MaxThrust
    c.thr_min_on_time = thr[0].MinOnTime
    c.base_density = 1.22
    c.scale_height = 8.0e3
    # 6U cubesat facets + two 1x2 m panels, Cd 2.2 (...Simulator.py:272-281)
    c.n_facets = len(FACETS)
    for i, (area, cd, normal, pos) in enumerate(FACETS):
        c.facet_area[i], c.facet_cd[i] = area, cd
        for k in range(3):
            c.facet_normal[i][k] = normal[k]
            c.facet_pos[i][k] = pos[k]
    return c


def config_to_dict(c):
    """Plain-Python view of a bsk_config (used by tests and the golden generator)."""
    out = {}
    for name, _ in BskConfig._fields_:
        v = getattr(c, name)
        if hasattr(v, "__len__"):
            v = np.ctypeslib.as_array(v).copy()
        out[name] = v
    return out
