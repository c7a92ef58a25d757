#!/usr/bin/env python3
"""Generate tests/golden/trajectories.json: 50-digit mpmath trajectories of the hot path.

An independent restatement (plain Python lists of mpf, no numpy, no C) of the same step
sequence the CPU oracle (oracle/bsk_oracle.c) and the HIP kernels execute: FSW chain every
``fsw_every`` ticks -> classic RK4 -> MRP shadow switch, then the observation.  Inputs (ICs and
config constants) are exact doubles; every derived constant and every operation is carried at
50 significant digits and only the final values are rounded to double.  The fp64
implementations must agree with these to ~1e-12 relative, far inside the 1e-9 budget of
BASELINE.json.

Run from the repo root:  python tests/golden/make_golden.py      (about two minutes)
No file under /root/reference is read: Basilisk itself is not available, so
this pins the restated equations, not Basilisk's binaries ("parity unpinned").
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from basilisk_env_amd._lib import GRAV_PM, GRAV_PM_J2, GRAV_SH, n_fields  # noqa: E402
from basilisk_env_amd.simulators.dynamics.gravity_sh import sh_index, synthetic_sh_coefficients  # noqa: E402
from basilisk_env_amd.simulators.dynamics.config import config_to_dict, default_config  # noqa: E402
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch  # noqa: E402

mp.mp.dps = 50
M = mp.mpf


# ------------------------------------------------------------------ vectors
def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def scale(s, a):
    return [s * x for x in a]


def add(a, b):
    return [x + y for x, y in zip(a, b)]


def sub(a, b):
    return [x - y for x, y in zip(a, b)]


def norm(a):
    return mp.sqrt(dot(a, a))


def matvec(m, v):
    return [dot(m[0], v), dot(m[1], v), dot(m[2], v)]


def inv3(m):
    return (mp.matrix(m) ** -1).tolist()


# ------------------------------------------------------------------ attitude kinematics
def mrp2c(q):
    q2 = dot(q, q)
    t = [[0, -q[2], q[1]], [q[2], 0, -q[0]], [-q[1], q[0], 0]]
    d = (1 + q2) ** 2
    C = [[M(int(i == j)) for j in range(3)] for i in range(3)]
    for i in range(3):
        for j in range(3):
            t2 = sum(t[i][k] * t[k][j] for k in range(3))
            C[i][j] += (8 * t2 - 4 * (1 - q2) * t[i][j]) / d
    return C


def c2mrp(C):
    tr = C[0][0] + C[1][1] + C[2][2]
    b2 = [(1 + tr) / 4, (1 + 2 * C[0][0] - tr) / 4, (1 + 2 * C[1][1] - tr) / 4, (1 + 2 * C[2][2] - tr) / 4]
    i = 0
    for j in range(1, 4):
        if b2[j] > b2[i]:
            i = j
    b = [None] * 4
    if i == 0:
        b[0] = mp.sqrt(b2[0])
        b[1] = (C[1][2] - C[2][1]) / 4 / b[0]
        b[2] = (C[2][0] - C[0][2]) / 4 / b[0]
        b[3] = (C[0][1] - C[1][0]) / 4 / b[0]
    elif i == 1:
        b[1] = mp.sqrt(b2[1])
        b[0] = (C[1][2] - C[2][1]) / 4 / b[1]
        if b[0] < 0:
            b[1], b[0] = -b[1], -b[0]
        b[2] = (C[0][1] + C[1][0]) / 4 / b[1]
        b[3] = (C[2][0] + C[0][2]) / 4 / b[1]
    elif i == 2:
        b[2] = mp.sqrt(b2[2])
        b[0] = (C[2][0] - C[0][2]) / 4 / b[2]
        if b[0] < 0:
            b[2], b[0] = -b[2], -b[0]
        b[1] = (C[0][1] + C[1][0]) / 4 / b[2]
        b[3] = (C[1][2] + C[2][1]) / 4 / b[2]
    else:
        b[3] = mp.sqrt(b2[3])
        b[0] = (C[0][1] - C[1][0]) / 4 / b[3]
        if b[0] < 0:
            b[3], b[0] = -b[3], -b[0]
        b[1] = (C[2][0] + C[0][2]) / 4 / b[3]
        b[2] = (C[1][2] + C[2][1]) / 4 / b[3]
    return [b[k + 1] / (1 + b[0]) for k in range(3)]


def submrp(q1, q2):
    s1 = list(q1)
    d1, d2 = dot(s1, s1), dot(q2, q2)
    den = 1 + d1 * d2 + 2 * dot(s1, q2)
    if abs(den) < M("0.1"):
        s1 = scale(-1 / d1, s1)
        d1 = dot(s1, s1)
        den = 1 + d1 * d2 + 2 * dot(s1, q2)
    t = cross(s1, q2)
    q = [((1 - d2) * s1[k] - (1 - d1) * q2[k] + 2 * t[k]) / den for k in range(3)]
    m = dot(q, q)
    if m > 1:
        q = scale(-1 / m, q)
    return q


# ------------------------------------------------------------------ model
class Model(object):
    def __init__(self, cfg):
        c = config_to_dict(cfg)
        self.n_rw = int(c["n_rw"])
        self.grav = int(c["gravity_model"])
        self.dt = M(float(c["dt"]))
        self.fsw_every = int(c["fsw_every"])
        self.fsw_lag = int(c["fsw_lag"])
        self.nav_lag = int(c["nav_lag"]) if int(c["n_rw"]) else 0
        self.mu, self.req, self.j2 = M(float(c["mu"])), M(float(c["req"])), M(float(c["j2"]))
        self.I = [[M(float(c["inertia"][3 * i + j])) for j in range(3)] for i in range(3)]
        self.gs = [[M(float(c["gs"][i][k])) for k in range(3)] for i in range(self.n_rw)]
        self.js = [M(float(c["js"][i])) for i in range(self.n_rw)]
        self.u_max, self.u_min, self.fc = M(float(c["u_max"])), M(float(c["u_min"])), M(float(c["f_coulomb"]))
        self.K, self.P = M(float(c["K"])), M(float(c["P"]))
        self.sR0N = [M(float(v)) for v in c["sigma_R0N"]]
        Cax = [[M(float(c["ctrl_axes"][3 * i + j])) for j in range(3)] for i in range(3)]
        D = [[self.I[a][b] - sum(self.js[i] * self.gs[i][a] * self.gs[i][b] for i in range(self.n_rw)) for b in range(3)]
             for a in range(3)]
        self.Dinv = inv3(D)
        if self.n_rw:
            cgs = [[dot(Cax[a], self.gs[i]) for i in range(self.n_rw)] for a in range(3)]
            Mi = inv3([[sum(cgs[a][i] * cgs[b][i] for i in range(self.n_rw)) for b in range(3)] for a in range(3)])
            self.map = []
            for i in range(self.n_rw):
                t = [sum(cgs[k][i] * Mi[k][a] for k in range(3)) for a in range(3)]
                self.map.append([sum(t[k] * Cax[k][b] for k in range(3)) for b in range(3)])
        self.wheel_limit, self.power_max = M(float(c["wheel_limit"])), M(float(c["power_max"]))
        self.reward_mult, self.failure_penalty = M(float(c["reward_mult"])), M(float(c["failure_penalty"]))
        self.r_min, self.max_length = M(float(c["r_min"])), int(c["max_length"])

    def set_scenario(self, cfg):
        """Power system, Sun third body and facet drag of the reference scenario,
        restated from the module documentation: conical eclipse with the two-disc lens area, cosine-law
        panel, Euler battery with clamping; mu_s [d/|d|^3 - s/|s|^3]; exponential atmosphere refreshed
        once per dyn tick, per-facet drag evaluated at every integrator stage."""
        from basilisk_env_amd._lib import FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        c = config_to_dict(cfg)
        fl = int(c["flags"])
        self.power, self.sun3, self.drag = bool(fl & FLAG_POWER), bool(fl & FLAG_SUN_THIRD_BODY), bool(fl & FLAG_DRAG)
        self.sun_r0 = [M(float(v)) for v in c["sun_r0"]]
        self.sun_v = [M(float(v)) for v in c["sun_v"]]
        self.mu_sun = M(float(c["mu_sun"]))
        self.nB = [M(float(v)) for v in c["panel_normal"]]
        self.panel = M(float(c["panel_area"])) * M(float(c["panel_efficiency"]))
        self.flux0, self.draw, self.cap = M(float(c["solar_flux"])), M(float(c["power_draw"])), M(float(c["storage_capacity"]))
        self.rho0, self.H, self.mass = M(float(c["base_density"])), M(float(c["scale_height"])), M(float(c["mass"]))
        self.facets = [(M(float(c["facet_area"][i])), M(float(c["facet_cd"][i])), [M(float(v)) for v in c["facet_normal"][i]],
                        [M(float(v)) for v in c["facet_pos"][i]]) for i in range(int(c["n_facets"]))]
        # rwDesatTask (row f2): thrMomentumManagement -> thrForceMapping -> thrMomentumDumping -> ideal thrusters
        from basilisk_env_amd._lib import FLAG_DESAT
        self.desat = bool(fl & FLAG_DESAT)
        self.n_thr = int(c["n_thr"])
        pos = [[M(float(v)) for v in c["thr_pos"][i]] for i in range(self.n_thr)]
        dirs = [[M(float(v)) for v in c["thr_dir"][i]] for i in range(self.n_thr)]
        self.fmax = M(float(c["thr_max_thrust"]))
        self.hs_min, self.min_fire, self.min_on = M(float(c["hs_min"])), M(float(c["thr_min_fire_time"])), M(float(c["thr_min_on_time"]))
        self.max_counter = int(c["thr_max_counter"])
        D = [cross(pos[i], dirs[i]) for i in range(self.n_thr)]                    # torque per unit thrust
        self.thr_f = [scale(self.fmax, dirs[i]) for i in range(self.n_thr)]
        self.thr_l = [scale(self.fmax, D[i]) for i in range(self.n_thr)]
        if self.desat:
            DDi = inv3([[sum(D[i][a] * D[i][b] for i in range(self.n_thr)) for b in range(3)] for a in range(3)])
            self.thr_map = [matvec(DDi, D[i]) for i in range(self.n_thr)]         # rows of D^T (D D^T)^-1

    power = sun3 = drag = desat = False

    def sun_at(self, tick):
        t = tick * self.dt
        return [self.sun_r0[k] + self.sun_v[k] * t for k in range(3)]

    def shadow(self, r, sun):
        RS = M(695000000)
        rHB = [sun[k] - r[k] for k in range(3)]
        nhp = norm(sun)
        if norm(rHB) < nhp:
            return M(1)
        f1, f2 = mp.asin((RS + self.req) / nhp), mp.asin((RS - self.req) / nhp)
        s, s0 = norm(r), -dot(r, sun) / nhp
        c1, c2 = s0 + self.req / mp.sin(f1), s0 - self.req / mp.sin(f2)
        l, l1, l2 = mp.sqrt(s * s - s0 * s0), c1 * mp.tan(f1), c2 * mp.tan(f2)
        if not (abs(l) < abs(l2) or abs(l) < abs(l1)):
            return M(1)
        nh, ns = norm(rHB), norm(r)
        a, b = mp.asin(RS / nh), mp.asin(self.req / ns)
        c = mp.acos(-dot(r, rHB) / (ns * nh))
        if c < b - a:
            return M(0)
        if c < a - b:
            return 1 - (b * b) / (a * a)
        if c < a + b:
            x = (c * c + a * a - b * b) / (2 * c)
            y = mp.sqrt(a * a - x * x)
            area = a * a * mp.acos(x / a) + b * b * mp.acos((c - x) / b) - c * y
            return 1 - area / (mp.pi * a * a)
        return M(1)

    def power_tick(self, x, sun, charge):
        sh = self.shadow(x[0:3], sun)
        d = [sun[k] - x[k] for k in range(3)]
        dm = norm(d)
        sB = matvec(mrp2c(x[6:9]), scale(1 / dm, d))
        proj = dot(self.nB, sB)
        if proj < 0:
            proj = M(0)
        AU = M(149597870700)
        p = self.flux0 * (AU / dm) ** 2 * proj * sh * self.panel + self.draw
        charge = charge + p * self.dt
        return min(max(charge, M(0)), self.cap), sh

    def drag_force(self, sig, vN, rho):
        BN = mrp2c(sig)
        vB = matvec(BN, vN)
        vm = norm(vB)
        vh = scale(1 / vm, vB)
        F, L = [M(0)] * 3, [M(0)] * 3
        for area, cd, n, pos in self.facets:
            proj = area * dot(n, vh)
            if proj > 0:
                f = scale(-vm * vm * cd * proj * rho / 2, vh)
                F, L = add(F, f), add(L, cross(pos, f))
        FN = [sum(BN[j][i] * F[j] for j in range(3)) for i in range(3)]     # [BN]^T F_B
        return scale(1 / self.mass, FN), L

    def desat_tick(self, env, x, first):
        Tc = self.fsw_every * self.dt
        if first:   # one momentum request per mode entry: dump what exceeds hs_min, on-pulsing minimum-norm impulses
            hs = [sum(self.js[i] * x[12 + i] * self.gs[i][k] for i in range(self.n_rw)) for k in range(3)]
            hm = norm(hs)
            dH = scale(-(hm - self.hs_min) / hm, hs) if hm > self.hs_min else [M(0)] * 3
            F = [dot(self.thr_map[i], dH) for i in range(self.n_thr)]
            fmin = min(F)
            env["thr_rem"] = [(F[i] - fmin) / self.fmax for i in range(self.n_thr)] + [M(0)] * (8 - self.n_thr)
            env["thr_cnt"] = 0
        if env["thr_cnt"] <= 0:   # fire: at most one control period per burst, short pulses dropped / stretched
            msg = list(env["thr_lim"])
            for i in range(self.n_thr):
                on = min(env["thr_rem"][i], Tc)
                if on < self.min_fire:
                    env["thr_rem"][i], msg[i] = M(0), M(0)
                    continue
                env["thr_rem"][i] -= on
                msg[i] = M(2 * self.fsw_every) if on >= Tc else mp.floor(max(on, self.min_on) * 2 / self.dt)
            env["thr_msg"] = msg            # thruster on-time command message; the thruster set reads it when it runs
            env["thr_cnt"] = self.max_counter
        else:
            env["thr_cnt"] -= 1

    def set_sh(self, degree, cbar, sbar, planet_rate):
        """Pines' normalised recursion, row-major tables as Basilisk documents them."""
        self.deg = degree
        self.C = {(l, m): M(float(cbar[sh_index(l, m)])) for l in range(degree + 1) for m in range(l + 1)}
        self.S = {(l, m): M(float(sbar[sh_index(l, m)])) for l in range(degree + 1) for m in range(l + 1)}
        self.planet_rate = M(float(planet_rate))
        K = lambda i: 1 if i == 0 else 2  # noqa: E731
        d = degree
        self.diag = {0: M(1)}
        for l in range(1, d + 2):
            self.diag[l] = mp.sqrt(M((2 * l + 1) * K(l)) / (2 * l * K(l - 1))) * self.diag[l - 1]
        self.n1, self.n2, self.nq1, self.nq2 = {}, {}, {}, {}
        for l in range(d + 2):
            for m in range(l + 1):
                if l >= m + 2:
                    self.n1[l, m] = mp.sqrt(M((2 * l + 1) * (2 * l - 1)) / ((l - m) * (l + m)))
                    self.n2[l, m] = mp.sqrt(M((l + m - 1) * (2 * l + 1) * (l - m - 1)) / ((l + m) * (l - m) * (2 * l - 3)))
        for l in range(d + 1):
            for m in range(l + 1):
                if m < l:
                    self.nq1[l, m] = mp.sqrt(M((l - m) * K(m) * (l + m + 1)) / K(m + 1))
                self.nq2[l, m] = mp.sqrt(M((l + m + 2) * (l + m + 1) * (2 * l + 1) * K(m)) / ((2 * l + 3) * K(m + 1)))
        self.kfac = K

    def sh_field(self, p):
        d, K = self.deg, self.kfac
        r = norm(p)
        s, t, u = p[0] / r, p[1] / r, p[2] / r
        A = {}
        for l in range(d + 2):
            A[l, l] = self.diag[l]
        for l in range(1, d + 2):
            A[l, l - 1] = mp.sqrt(M(2 * l * K(l - 1)) / K(l)) * A[l, l] * u
        rE, iM = [M(1)], [M(0)]
        for m in range(d + 2):
            for l in range(m + 2, d + 2):
                A[l, m] = u * self.n1[l, m] * A[l - 1, m] - self.n2[l, m] * A[l - 2, m]
            if m > 0:
                rE.append(s * rE[m - 1] - t * iM[m - 1])
                iM.append(s * iM[m - 1] + t * rE[m - 1])
        rho = self.req / r
        rhol = [self.mu / r, self.mu / r * rho]
        a1 = a2 = a3 = M(0)
        a4 = -rhol[1] / self.req
        for l in range(1, d + 1):
            rhol.append(rho * rhol[l])
            s1 = s2 = s3 = s4 = M(0)
            for m in range(l + 1):
                cb, sb = self.C[l, m], self.S[l, m]
                D = cb * rE[m] + sb * iM[m]
                E = cb * rE[m - 1] + sb * iM[m - 1] if m > 0 else M(0)
                F = sb * rE[m - 1] - cb * iM[m - 1] if m > 0 else M(0)
                s1 += m * A[l, m] * E
                s2 += m * A[l, m] * F
                if m < l:
                    s3 += self.nq1[l, m] * A[l, m + 1] * D
                s4 += self.nq2[l, m] * A[l + 1, m + 1] * D
            w = rhol[l + 1] / self.req
            a1 += w * s1
            a2 += w * s2
            a3 += w * s3
            a4 -= w * s4
        return [a1 + s * a4, a2 + t * a4, a3 + u * a4]

    def gravity(self, r, t=None):
        if self.grav == GRAV_SH:
            th = self.planet_rate * t
            ct, st = mp.cos(th), mp.sin(th)
            p = [ct * r[0] + st * r[1], -st * r[0] + ct * r[1], r[2]]
            ap = self.sh_field(p)
            return [ct * ap[0] - st * ap[1], st * ap[0] + ct * ap[1], ap[2]]
        rm = norm(r)
        a = scale(-self.mu / rm ** 3, r)
        if self.grav == GRAV_PM_J2:
            z2 = (r[2] / rm) ** 2
            k = M(3) / 2 * self.j2 * self.mu * self.req ** 2 / rm ** 5
            a = [a[0] + k * r[0] * (5 * z2 - 1), a[1] + k * r[1] * (5 * z2 - 1), a[2] + k * r[2] * (5 * z2 - 3)]
        return a

    def wheel_torque(self, x, u):
        """motor torque + Coulomb friction from the wheel speeds at the start of the step"""
        tq = []
        for i in range(self.n_rw):
            Om = x[12 + i]
            fr = -self.fc if Om > 0 else (self.fc if Om < 0 else M(0))
            tq.append(u[i] + fr)
        return tq

    def eom(self, x, tq, lext, t=None, sun=None, rho=None, thr=None):
        r, v, s, w, Om = x[0:3], x[3:6], x[6:9], x[9:12], x[12:]
        dv = self.gravity(r, t)
        if self.sun3:
            d = [sun[k] - r[k] for k in range(3)]
            dv = add(dv, scale(self.mu_sun, add(scale(1 / norm(d) ** 3, d), scale(-1 / norm(sun) ** 3, sun))))
        if self.drag:
            aN, LB = self.drag_force(s, v, rho)
            dv, lext = add(dv, aN), add(lext, LB)
        if thr is not None:
            lim, e2 = thr
            FB = [M(0)] * 3
            for i in range(self.n_thr):
                if lim[i] > 0 and e2 <= lim[i]:      # inside its burst (time counted in half dyn steps)
                    FB, lext = add(FB, self.thr_f[i]), add(lext, self.thr_l[i])
            BN = mrp2c(s)
            dv = add(dv, scale(1 / self.mass, [sum(BN[j][i] * FB[j] for j in range(3)) for i in range(3)]))
        s2, sw, sxw = dot(s, s), dot(s, w), cross(s, w)
        ds = [((1 - s2) * w[k] + 2 * sxw[k] + 2 * sw * s[k]) / 4 for k in range(3)]
        rhs = add(scale(-1, cross(w, matvec(self.I, w))), lext)
        for i in range(self.n_rw):
            wg = cross(w, self.gs[i])
            rhs = [rhs[k] - self.gs[i][k] * tq[i] - self.js[i] * Om[i] * wg[k] for k in range(3)]
        dw = matvec(self.Dinv, rhs)
        dOm = [tq[i] / self.js[i] - dot(self.gs[i], dw) for i in range(self.n_rw)]
        return list(v) + dv + ds + dw + dOm

    def rk4(self, x, u, lext, t=None, sun=None, thr=None):
        h = self.dt
        ax = lambda a, k, y: [yi + a * ki for yi, ki in zip(y, k)]  # noqa: E731
        u = self.wheel_torque(x, u)   # held over the four stages
        t = M(0) if t is None else t
        rho = self.rho0 * mp.exp(-(norm(x[0:3]) - self.req) / self.H) if self.drag else None   # once per dyn tick
        th = (lambda de: None) if thr is None else (lambda de: (thr[0], thr[1] + de))
        k1 = self.eom(x, u, lext, t, sun, rho, th(0))
        k2 = self.eom(ax(h / 2, k1, x), u, lext, t + h / 2, sun, rho, th(1))
        k3 = self.eom(ax(h / 2, k2, x), u, lext, t + h / 2, sun, rho, th(1))
        k4 = self.eom(ax(h, k3, x), u, lext, t + h, sun, rho, th(2))
        x = [x[i] + h / 6 * k1[i] + h / 3 * k2[i] + h / 3 * k3[i] + h / 6 * k4[i] for i in range(len(x))]
        s2 = dot(x[6:9], x[6:9])
        if s2 > 1:
            x[6:9] = scale(-1 / s2, x[6:9])
        return x

    def guidance(self, x, action):
        r, v, s, w = x[0:3], x[3:6], x[6:9], x[9:12]
        if action == 0 and norm(r) == 0:
            # hillPoint on a navigation message nobody has written yet: zero unit vectors, zero DCM -> zero MRP,
            # rates zeroed by the module's radius guard
            sRN, wRN, dwRN = [M(0)] * 3, [M(0)] * 3, [M(0)] * 3
        elif action == 0:
            rm = norm(r)
            h = cross(r, v)
            hm = norm(h)
            e_r, e_h = scale(1 / rm, r), scale(1 / hm, h)
            e_t = cross(e_h, e_r)
            sRN = c2mrp([e_r, e_t, e_h])
            dfdt = hm / rm ** 2
            ddf = -2 * dot(v, e_r) / rm * dfdt
            wRN, dwRN = scale(dfdt, e_h), scale(ddf, e_h)
        else:
            sRN, wRN, dwRN = list(self.sR0N), [M(0)] * 3, [M(0)] * 3
        sBR = submrp(s, sRN)
        BN = mrp2c(s)
        wRN_B, dwRN_B = matvec(BN, wRN), matvec(BN, dwRN)
        return sBR, sub(w, wRN_B), wRN_B, dwRN_B

    def control(self, g):
        sBR, wBR, wRN, dwRN = g
        wBN = add(wBR, wRN)
        Lr = add(scale(self.K, sBR), scale(self.P, wBR))
        Lr = sub(Lr, cross(wRN, matvec(self.I, wBN)))
        Lr = add(Lr, matvec(self.I, sub(cross(wBN, wRN), dwRN)))
        Lr = scale(-1, Lr)
        u = []
        for i in range(self.n_rw):
            us = -dot(self.map[i], Lr)
            if self.u_max > 0:
                us = min(max(us, -self.u_max), self.u_max)
            if abs(us) < self.u_min:
                us = M(0)
            u.append(us)
        return u

    def step(self, env, action, substeps):
        """env: dict(x, u, lext, charge, steps, ticks) — one spacecraft; mutated."""
        x, u = env["x"], env["u"]
        sun = self.sun_at(env["ticks"]) if (self.power or self.sun3) else None   # held over the env step
        shadow = M(1)
        state = {"first": True}

        def fsw_tick(nav):
            """every enabled FSW task once, in priority / insertion order, on the navigation message `nav`"""
            g = self.guidance(nav, action)              # hillPoint | inertial3D (priority 100) ... attTrackingError
            env["sbr"] = norm(g[0])
            if self.fsw_lag:
                # mrpControlTask: MRP_Feedback was added BEFORE attTrackingError: it reads the att_guidance message as
                # the previous FSW tick left it (all zeros = never written), then this tick's tracking error overwrites it
                env["cmd_msg"] = self.control(env["guid"])
                env["guid"] = g
            else:
                env["cmd_msg"] = self.control(g)
            if self.desat and action == 2:
                self.desat_tick(env, nav, state["first"])
            state["first"] = False

        def latch(u):
            """the dynamics task's effectors read the FSW output messages after integrating to the current time"""
            if env.get("cmd_msg") is not None:
                u = env.pop("cmd_msg")
            if env.get("thr_msg") is not None:
                env["thr_lim"], env["thr_t0"] = env.pop("thr_msg"), env["ticks"]
            return u

        if self.nav_lag and env["ticks"] == 0:
            # FSW priorities 100 / 50 against the dynamics tasks' default: the FSW tasks of t = 0 run before any
            # dynamics task has written a message (zeros)
            fsw_tick([M(0)] * len(x))
            u = latch(u)
        for _ in range(substeps):
            if self.n_rw and not self.nav_lag and env["ticks"] % self.fsw_every == 0:
                fsw_tick(x)
                u = latch(u)
            if self.nav_lag and (env["ticks"] + 1) % self.fsw_every == 0:
                fsw_tick(x)      # the FSW tasks of the NEXT time run before the dynamics task integrates to it
            thr = None
            if self.desat:
                e2 = 2 * (env["ticks"] - env["thr_t0"])
                if any(l > 0 and e2 <= l for l in env["thr_lim"]):
                    thr = (env["thr_lim"], e2)
            x = self.rk4(x, u, env["lext"], env["ticks"] * self.dt, sun, thr)
            env["ticks"] += 1
            if self.nav_lag:
                u = latch(u)
            if self.power:
                env["charge"], shadow = self.power_tick(x, sun, env["charge"])
        env["x"], env["u"] = x, u
        # obs[0]: the logged att_guidance message — the last FSW tick's with nav_lag, else the end state's tracking error
        o0 = env["sbr"] if self.nav_lag else norm(self.guidance(x, action)[0])
        o1 = norm(x[9:12])
        o2 = mp.sqrt(sum(v * v for v in x[12:])) / self.wheel_limit if self.n_rw else M(0)
        o3 = env["charge"] / 3600 / self.power_max
        why = 0
        rew = self.reward_mult / (1 + o0 * o0) if action == 0 else M(0)
        if env["steps"] >= self.max_length:
            why |= 1
        if o2 > 1:
            why |= 2
            rew -= self.failure_penalty
        if o3 == 0:
            why |= 4
            rew -= self.failure_penalty
        if norm(x[0:3]) < self.r_min:
            why |= 8
        env["steps"] += 1
        return [o0, o1, o2, o3, shadow], rew, why


def run_case(name, n_rw, grav, n_envs, seed, schedule, cfg_edit=None, sh=None, ic_edit=None):
    cfg = default_config(n_rw=n_rw, gravity_model=grav)
    if cfg_edit:
        cfg_edit(cfg)
    model = Model(cfg)
    model.set_scenario(cfg)
    if sh is not None:
        model.set_sh(sh[0], sh[1], sh[2], cfg.planet_rate)
    ic = sample_ic_batch(n_envs, n_rw, seed=seed)
    if ic_edit:
        ic_edit(cfg, ic)
    nf = n_fields(n_rw)
    t = 12 + n_rw
    envs = []
    for e in range(n_envs):
        envs.append({"x": [M(float(ic[f, e])) for f in range(12 + n_rw)], "u": [M(0)] * n_rw,
                     "lext": [M(float(ic[t + k, e])) for k in range(3)], "charge": M(float(ic[t + 7, e])),
                     "steps": 0, "ticks": 0, "thr_rem": [M(0)] * 8, "thr_lim": [M(0)] * 8, "thr_t0": 0, "thr_cnt": 0,
                     "guid": ([M(0)] * 3, [M(0)] * 3, [M(0)] * 3, [M(0)] * 3), "sbr": M(0)})   # att_guidance message, never written yet
    calls = []
    for ci, (actions, substeps) in enumerate(schedule):
        obs, rews, whys = [], [], []
        for e, env in enumerate(envs):
            o, r, w = model.step(env, int(actions[e]), substeps)
            obs.append([float(v) for v in o])
            rews.append(float(r))
            whys.append(w)
        state = np.zeros((nf, n_envs))
        for e, env in enumerate(envs):
            state[:12 + n_rw, e] = [float(v) for v in env["x"]]
            state[t:t + 3, e] = [float(v) for v in env["lext"]]
            state[t + 3:t + 3 + n_rw, e] = [float(v) for v in env["u"]]
            state[t + 7, e] = float(env["charge"])
            if model.fsw_lag and n_rw:
                # the slab keeps the torque the held message maps to (BSK_T_UPEND), not the message itself
                state[t + 26:t + 26 + n_rw, e] = [float(v) for v in model.control(env["guid"])]
            state[t + 30, e] = float(env["sbr"])
            if model.desat:
                state[t + 8:t + 16, e] = [float(v) for v in env["thr_rem"]]
                state[t + 16:t + 24, e] = [float(v) for v in env["thr_lim"]]
                state[t + 24, e], state[t + 25, e] = env["thr_t0"], env["thr_cnt"]
        calls.append({"actions": [int(a) for a in actions], "substeps": int(substeps), "state": state.tolist(),
                      "obs": np.array(obs).T.tolist(), "reward": rews, "reason": whys})
        print("  %s call %d/%d done" % (name, ci + 1, len(schedule)), flush=True)
    out = {"name": name, "n_rw": n_rw, "gravity_model": grav, "n_envs": n_envs, "seed": seed, "ic": ic.tolist(),
           "calls": calls}
    if sh is not None:
        out["sh_degree"], out["cbar"], out["sbar"] = sh[0], [float(v) for v in sh[1]], [float(v) for v in sh[2]]
    if cfg_edit and sh is None:
        out["cfg_edit"] = cfg_edit.__name__
    return out


def main():
    """No arguments: regenerate every case.  `--only NAME [NAME..]`: regenerate just those cases and merge
    them into the existing trajectories.json (the others are kept byte for byte)."""
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    n = 8
    rng = np.random.Generator(np.random.PCG64(99))
    sched = [(rng.integers(0, 3, n), int(k)) for k in (50, 37, 3, 110, 50, 50, 25, 75)]
    # config-5 shape at low degree: degree-8 synthetic field (harmonics exaggerated x100 so that a wrong
    # term shows), rotating planet, 3 wheels, one mode per env; 4 envs, checkpoints at 1, 10, 100, 400 steps
    cb, sb = synthetic_sh_coefficients(8, seed=8)
    cb[3:] *= 100.0
    sb[3:] *= 100.0

    def sh_edit(cfg):
        cfg.sh_degree = 8

    def scenario_edit(cfg):
        """full scenario minus desaturation, with an atmosphere dense enough at 500 km for drag to show"""
        from basilisk_env_amd._lib import FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        cfg.base_density, cfg.scale_height = 1e-9, 100e3

    def nolag_edit(cfg):
        """guidance and control on the same FSW tick, on the state of that tick (fsw_lag = nav_lag = 0)"""
        cfg.fsw_lag = 0
        cfg.nav_lag = 0

    def navnow_edit(cfg):
        """the reference's model order inside mrpControlTask, but FSW ticks on the state of their own time (nav_lag = 0)"""
        cfg.nav_lag = 0

    def desat_edit(cfg):
        from basilisk_env_amd._lib import FLAG_DESAT, FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG | FLAG_DESAT

    def dense_drag_edit(cfg):
        """1 s dyn ticks under a thin-scale-height atmosphere: strong drag at 200 km, and the density's exponent moves by up to
        0.02 per tick - both sides of the kernels' increment guard (bsk_device.hpp: Atmo; the oracle and this generator evaluate
        the exponential every tick)"""
        from basilisk_env_amd._lib import FLAG_DRAG, FLAG_POWER, FLAG_SUN_THIRD_BODY
        cfg.flags |= FLAG_POWER | FLAG_SUN_THIRD_BODY | FLAG_DRAG
        cfg.dt = 1.0
        cfg.base_density, cfg.scale_height = 1e-4, 20e3

    def full_inertia_edit(cfg):
        """a GENERAL hub - symmetric positive-definite inertia matrix with products of inertia, one wheel axis tilted by nine
        degrees - in the full scenario minus desaturation: the case that pins the step kernels' general-inertia family
        (3 x 3 back-substitution, full W = sum Js g g^T; csrc/bsk_capi.hip picks it whenever an off-diagonal is non-zero)"""
        scenario_edit(cfg)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import general_hub       # (an edit of the INPUT constants, shared with the tests that rebuild the config)
        general_hub(cfg)

    def low_perigee_ic(cfg, ic):
        """envs 0..3 at 180 - 260 km with radial velocities of +-380 / +-250 m/s (the others keep their sampled 500 km orbits)"""
        for e, (alt, vr, ang) in enumerate(((180e3, 380.0, 0.3), (220e3, -380.0, 1.7), (260e3, 250.0, 3.1), (200e3, -250.0, 4.9))):
            rhat = np.array([np.cos(ang), np.sin(ang) * 0.8, np.sin(ang) * 0.6])
            rhat /= np.linalg.norm(rhat)
            that = np.cross([0.3, -0.5, 0.8], rhat)
            that /= np.linalg.norm(that)
            rm = cfg.req + alt
            ic[0:3, e] = rm * rhat
            ic[3:6, e] = np.sqrt(cfg.mu / rm) * that + vr * rhat

    def penumbra_ic(cfg, ic):
        """envs 0..3 fly through the penumbra band behind the Earth during the run (the others keep their
        sampled orbits); env 3 starts with a nearly empty battery"""
        sun = np.array(cfg.sun_r0)
        shat = sun / np.linalg.norm(sun)
        perp = np.cross(shat, [0.0, 0.0, 1.0])
        perp /= np.linalg.norm(perp)
        for e, off in enumerate((-20e3, 5e3, 30e3, 45e3)):
            r = -7000e3 * shat + (cfg.req + off) * perp
            v = -7400.0 * perp * (1 if e % 2 == 0 else -1) + 300.0 * np.cross(shat, perp)
            ic[0:3, e], ic[3:6, e] = r, v
        ic[12 + cfg.n_rw + 7, 3] = 3.0

    recipes = [
        # config-2 shape: point mass + MRP attitude, no wheels; checkpoints at 1, 10, 100, 1000 RK4 steps
        ("pm_norw", lambda: run_case("pm_norw", 0, GRAV_PM, n, 11, [(np.zeros(n, int), k) for k in (1, 9, 90, 900)])),
        # config-3 shape: J2 + 4-wheel pyramid; nadir/sun-point per env; u is held across the calls
        ("j2_rw4", lambda: run_case("j2_rw4", 4, GRAV_PM_J2, n, 12, [(np.arange(n) % 2, k) for k in (1, 9, 90, 900)])),
        # the same with fsw_lag = 0 (first three checkpoints)
        ("j2_rw4_nolag", lambda: run_case("j2_rw4_nolag", 4, GRAV_PM_J2, n, 12, [(np.arange(n) % 2, k) for k in (1, 9, 90)],
                                          cfg_edit=nolag_edit)),
        ("j2_rw4_navnow", lambda: run_case("j2_rw4_navnow", 4, GRAV_PM_J2, n, 12, [(np.arange(n) % 2, k) for k in (1, 9, 90)],
                                           cfg_edit=navnow_edit)),
        # reference wiring: point mass + 3-wheel triad, mode switches between calls, odd call lengths
        ("pm_rw3_modes", lambda: run_case("pm_rw3_modes", 3, GRAV_PM, n, 13, sched)),
        # rows f1 / f3: power system (with penumbra crossings), Sun third body, facet drag; J2 + 3 wheels
        ("scenario_rw3", lambda: run_case("scenario_rw3", 3, GRAV_PM_J2, 6, 15, [(np.array([0, 1, 0, 1, 2, 0]), k) for k in (1, 49, 150, 200)],
                                          cfg_edit=scenario_edit, ic_edit=penumbra_ic)),
        # row f2: desaturation (action 2 on envs whose wheel momentum exceeds hs_min), full scenario otherwise
        ("desat_rw3", lambda: run_case("desat_rw3", 3, GRAV_PM, 6, 16, [(np.array([2, 2, 0, 2, 1, 2]), k) for k in (7, 43, 100, 150)],
                                       cfg_edit=desat_edit)),
        # the density along a fast radial motion (round 4: the kernels advance it incrementally): J2, 1 s ticks, no wheels (with
        # them the reference's gains at a 10 s control period are an unstable loop that amplifies rounding to 5e-9 in 300 steps)
        ("drag_dt1_norw", lambda: run_case("drag_dt1_norw", 0, GRAV_PM_J2, 6, 17, [(np.array([0, 1, 0, 1, 0, 0]), k) for k in (1, 9, 90, 200)],
                                           cfg_edit=dense_drag_edit, ic_edit=low_perigee_ic)),
        ("full_inertia_rw4", lambda: run_case("full_inertia_rw4", 4, GRAV_PM_J2, 6, 18, [(np.array([0, 1, 0, 1, 2, 0]), k) for k in (1, 9, 90, 200)],
                                              cfg_edit=full_inertia_edit)),
        ("sh8_rw3", lambda: run_case("sh8_rw3", 3, GRAV_SH, 4, 14, [(np.array([0, 1, 2, 0]), k) for k in (1, 9, 90, 300)],
                                     cfg_edit=sh_edit, sh=(8, cb, sb))),
    ]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trajectories.json")
    cases = []
    if only is not None:
        with open(out) as f:
            cases = json.load(f)["cases"]
    for name, make in recipes:
        if only is not None and name not in only:
            continue
        case = make()
        idx = [i for i, c in enumerate(cases) if c["name"] == name]
        if idx:
            cases[idx[0]] = case
        else:
            cases.append(case)
    with open(out, "w") as f:
        json.dump({"dps": mp.mp.dps, "generator": "tests/golden/make_golden.py", "cases": cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
