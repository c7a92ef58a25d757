/*
 * bsk_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (basilisk_env_amd + libbskgpu.so) never links, imports or calls it.
 *
 * PARITY UNPINNED.  The arithmetic of the hot path lives in the third-party Basilisk engine
 * (AVS Lab, CU Boulder; imported at reference basilisk_env/simulators/leoPowerAttitudeSimulator.py:5-27).
 * Basilisk is neither vendored under /root/reference nor pinned to a version (setup.py:5,
 * README.md:11; API generation 1.x by call-site names such as spacecraftPlus, :213), and the
 * reference has no tests or golden vectors for this path.  This file therefore restates the
 * algorithm Basilisk publishes for each module the reference wires up, anchored on the
 * reference's call sites (cited per function below), and is itself pinned by
 *   (a) a 50-digit mpmath re-implementation of the same step sequence (tests/golden/make_golden.py),
 *   (b) closed-form / conservation known-answer tests (tests/test_oracle_kat.py),
 *   (c) the one reference module importable without Basilisk (initial_conditions/sc_attitudes.py)
 *       for the initial-condition sampler (tests/golden/ic_random_tumble.json).
 *
 * Style: deliberately plain — one spacecraft at a time, 3-vectors and 3x3 matrices through
 * small helpers, no fused arithmetic (built with -ffp-contract=off), so that it is an
 * independent check of the hand-unrolled HIP kernels rather than a copy of them.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bskgpu.h"

/* ---------------------------------------------------------------- small algebra */
static void v3set(double a, double b, double c, double o[3]) { o[0] = a; o[1] = b; o[2] = c; }
static void v3copy(const double a[3], double o[3]) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static double v3dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3norm(const double a[3]) { return sqrt(v3dot(a, a)); }
static void v3scale(double s, const double a[3], double o[3]) { o[0] = s * a[0]; o[1] = s * a[1]; o[2] = s * a[2]; }
static void v3add(const double a[3], const double b[3], double o[3]) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static void v3sub(const double a[3], const double b[3], double o[3]) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void v3cross(const double a[3], const double b[3], double o[3]) {
    double t[3];
    t[0] = a[1] * b[2] - a[2] * b[1];
    t[1] = a[2] * b[0] - a[0] * b[2];
    t[2] = a[0] * b[1] - a[1] * b[0];
    v3copy(t, o);
}
static void m33v3(const double m[9], const double v[3], double o[3]) {
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = m[3 * i] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
    v3copy(t, o);
}
/* Engine behaviours this restatement had to DECIDE (each [BSK-recall]).  The defaults (0) are what the
   kernels implement; the alternatives exist so that every decision's weight on a trajectory is a measured number
   (tests/test_oracle_decisions.py) and can be flipped the day a Basilisk build settles it (orc_set_decision). */
static int g_friction_per_stage = 0;  /* 1: Coulomb friction re-evaluated from the stage state inside the equations of motion */
static int g_sun_per_tick = 0;        /* 1: Sun position advanced every dyn tick instead of held over the env step (SPICE task rate) */
static int g_t0_real_messages = 0;    /* 1: the FSW tick at t = 0 (nav_lag) reads the initial state instead of unwritten (zero) messages */
static int m33inv(const double m[9], double o[9]) {
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0) return -1;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 0;
}

/* ---------------------------------------------------------------- MRP kinematics
 * Basilisk utilities/rigidBodyKinematics (used by hillPoint, attTrackingError; the reference
 * imports it as rbk, leoPowerAttitudeSimulator.py:16). */

/* MRP -> DCM:  C = I + (8 s~^2 - 4 (1-s^2) s~) / (1+s^2)^2 */
static void mrp2c(const double q[3], double c[9]) {
    double q2 = v3dot(q, q), d = (1.0 + q2) * (1.0 + q2), a = 8.0 / d, b = 4.0 * (1.0 - q2) / d;
    double t[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
    double t2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += t[3 * i + k] * t[3 * k + j];
            t2[3 * i + j] = s;
        }
    for (int i = 0; i < 9; ++i) c[i] = a * t2[i] - b * t[i];
    c[0] += 1.0; c[4] += 1.0; c[8] += 1.0;
}

/* DCM -> Euler parameters by Sheppard's method, b0 >= 0; then MRP = b(1:3)/(1+b0). */
static void c2mrp(const double c[9], double q[3]) {
    double tr = c[0] + c[4] + c[8];
    double b2[4] = {(1 + tr) / 4., (1 + 2 * c[0] - tr) / 4., (1 + 2 * c[4] - tr) / 4., (1 + 2 * c[8] - tr) / 4.};
    int i = 0;
    for (int j = 1; j < 4; ++j) if (b2[j] > b2[i]) i = j;
    double b[4];
    switch (i) {
    case 0:
        b[0] = sqrt(b2[0]);
        b[1] = (c[5] - c[7]) / 4. / b[0]; b[2] = (c[6] - c[2]) / 4. / b[0]; b[3] = (c[1] - c[3]) / 4. / b[0];
        break;
    case 1:
        b[1] = sqrt(b2[1]);
        b[0] = (c[5] - c[7]) / 4. / b[1];
        if (b[0] < 0) { b[1] = -b[1]; b[0] = -b[0]; }
        b[2] = (c[1] + c[3]) / 4. / b[1]; b[3] = (c[6] + c[2]) / 4. / b[1];
        break;
    case 2:
        b[2] = sqrt(b2[2]);
        b[0] = (c[6] - c[2]) / 4. / b[2];
        if (b[0] < 0) { b[2] = -b[2]; b[0] = -b[0]; }
        b[1] = (c[1] + c[3]) / 4. / b[2]; b[3] = (c[5] + c[7]) / 4. / b[2];
        break;
    default:
        b[3] = sqrt(b2[3]);
        b[0] = (c[1] - c[3]) / 4. / b[3];
        if (b[0] < 0) { b[3] = -b[3]; b[0] = -b[0]; }
        b[1] = (c[6] + c[2]) / 4. / b[3]; b[2] = (c[5] + c[7]) / 4. / b[3];
        break;
    }
    for (int k = 0; k < 3; ++k) q[k] = b[k + 1] / (1.0 + b[0]);
}

/* q = q1 (-) q2 : relative MRP, with the near-singular guard (|den| < 0.1 -> use the shadow of
 * q1) and the final map to the inner set |q| <= 1. */
static void submrp(const double q1in[3], const double q2[3], double q[3]) {
    double s1[3], t[3];
    v3copy(q1in, s1);
    double d1 = v3dot(s1, s1), d2 = v3dot(q2, q2);
    double den = 1.0 + d1 * d2 + 2.0 * v3dot(s1, q2);
    if (fabs(den) < 0.1) {
        v3scale(-1.0 / d1, s1, s1);
        d1 = v3dot(s1, s1);
        den = 1.0 + d1 * d2 + 2.0 * v3dot(s1, q2);
    }
    v3cross(s1, q2, t);
    for (int k = 0; k < 3; ++k) q[k] = ((1.0 - d2) * s1[k] - (1.0 - d1) * q2[k] + 2.0 * t[k]) / den;
    double m = v3dot(q, q);
    if (m > 1.0) v3scale(-1.0 / m, q, q);
}

/* ---------------------------------------------------------------- derived constants */
typedef struct {
    const bsk_config* c;
    double dinv[9];                /* (I_sc - sum Js g g^T)^-1 : hub back-substitution matrix  */
    double map[BSK_MAX_RW][3];     /* rwMotorTorque: [CGs]^T ([CGs][CGs]^T)^-1 [C]             */
    /* spherical harmonics (Pines) */
    int deg;
    const double *cbar, *sbar;     /* packed l(l+1)/2+m                                        */
    double *abar, *n1, *n2, *nq1, *nq2; /* (deg+2)x(deg+2) square, row l col m                 */
    /* thrusters */
    double thr_map[BSK_MAX_THR][3];  /* thrForceMapping: [D]^T ([D][D]^T)^-1, D_i = r_i x dir_i */
    double thr_f[BSK_MAX_THR][3], thr_l[BSK_MAX_THR][3]; /* force / torque of thruster i at full thrust */
} orc_ctx;

#define SQ(l, m) ((l) * (ctx->deg + 2) + (m))
static double kfac(int i) { return i == 0 ? 1.0 : 2.0; }

static void sh_init(orc_ctx* ctx) {
    int d = ctx->deg, w = d + 2;
    size_t n = (size_t)w * w;
    ctx->abar = calloc(n, sizeof(double)); ctx->n1 = calloc(n, sizeof(double)); ctx->n2 = calloc(n, sizeof(double));
    ctx->nq1 = calloc(n, sizeof(double)); ctx->nq2 = calloc(n, sizeof(double));
    for (int l = 0; l <= d + 1; ++l) {
        ctx->abar[SQ(l, l)] = (l == 0) ? 1.0
            : sqrt((double)(2 * l + 1) * kfac(l) / ((double)(2 * l) * kfac(l - 1))) * ctx->abar[SQ(l - 1, l - 1)];
        for (int m = 0; m <= l; ++m)
            if (l >= m + 2) {
                ctx->n1[SQ(l, m)] = sqrt((double)(2 * l + 1) * (double)(2 * l - 1) / ((double)(l - m) * (double)(l + m)));
                ctx->n2[SQ(l, m)] = sqrt((double)(l + m - 1) * (double)(2 * l + 1) * (double)(l - m - 1) /
                                         ((double)(l + m) * (double)(l - m) * (double)(2 * l - 3)));
            }
    }
    for (int l = 0; l <= d; ++l)
        for (int m = 0; m <= l; ++m) {
            if (m < l) ctx->nq1[SQ(l, m)] = sqrt((double)(l - m) * kfac(m) * (double)(l + m + 1) / kfac(m + 1));
            ctx->nq2[SQ(l, m)] = sqrt((double)(l + m + 2) * (double)(l + m + 1) * (double)(2 * l + 1) * kfac(m) /
                                      ((double)(2 * l + 3) * kfac(m + 1)));
        }
}

/* Pines' normalised, singularity-free spherical-harmonic gravity as Basilisk's gravityEffector
 * documents it (SURVEY.md §8 note N1; harmonics hook at reference
 * opNav_models/BSK_OpNavDynamics.py:211-214).  pos in the planet-fixed frame; includes degree 0. */
static void sh_field(orc_ctx* ctx, const double pos[3], double acc[3]) {
    const bsk_config* c = ctx->c;
    int d = ctx->deg;
    double r = v3norm(pos), s = pos[0] / r, t = pos[1] / r, u = pos[2] / r;
    /* per-call working copy: the diagonal is constant, the rest is filled below */
    double A[(BSK_MAX_SH_DEGREE + 2) * (BSK_MAX_SH_DEGREE + 2)];
    for (int l = 0; l <= d + 1; ++l) A[SQ(l, l)] = ctx->abar[SQ(l, l)];
    double rE[BSK_MAX_SH_DEGREE + 2], iM[BSK_MAX_SH_DEGREE + 2], rhol[BSK_MAX_SH_DEGREE + 3];
    for (int l = 1; l <= d + 1; ++l)
        A[SQ(l, l - 1)] = sqrt((double)(2 * l) * kfac(l - 1) / kfac(l)) * A[SQ(l, l)] * u;
    for (int m = 0; m <= d + 1; ++m) {
        for (int l = m + 2; l <= d + 1; ++l)
            A[SQ(l, m)] = u * ctx->n1[SQ(l, m)] * A[SQ(l - 1, m)] - ctx->n2[SQ(l, m)] * A[SQ(l - 2, m)];
        if (m == 0) { rE[0] = 1.0; iM[0] = 0.0; }
        else { rE[m] = s * rE[m - 1] - t * iM[m - 1]; iM[m] = s * iM[m - 1] + t * rE[m - 1]; }
    }
    double rho = c->req / r;
    rhol[0] = c->mu / r; rhol[1] = rhol[0] * rho;
    double a1 = 0, a2 = 0, a3 = 0, a4 = -rhol[1] / c->req;
    for (int l = 1; l <= d; ++l) {
        rhol[l + 1] = rho * rhol[l];
        double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
        for (int m = 0; m <= l; ++m) {
            double cb = ctx->cbar[l * (l + 1) / 2 + m], sb = ctx->sbar[l * (l + 1) / 2 + m];
            double D = cb * rE[m] + sb * iM[m];
            double E = (m == 0) ? 0.0 : cb * rE[m - 1] + sb * iM[m - 1];
            double F = (m == 0) ? 0.0 : sb * rE[m - 1] - cb * iM[m - 1];
            s1 += m * A[SQ(l, m)] * E;
            s2 += m * A[SQ(l, m)] * F;
            if (m < l) s3 += ctx->nq1[SQ(l, m)] * A[SQ(l, m + 1)] * D;
            s4 += ctx->nq2[SQ(l, m)] * A[SQ(l + 1, m + 1)] * D;
        }
        double w = rhol[l + 1] / c->req;
        a1 += w * s1; a2 += w * s2; a3 += w * s3; a4 -= w * s4;
    }
    acc[0] = a1 + s * a4; acc[1] = a2 + t * a4; acc[2] = a3 + u * a4;
}

/* ---------------------------------------------------------------- equations of motion
 * State x = [r(3) v(3) sigma(3) omega(3) Omega(n_rw)], spacecraftPlus hub + gravityEffector +
 * reactionWheelStateEffector (balanced wheels) + extForceTorque as composed at reference
 * leoPowerAttitudeSimulator.py:213-232 (spacecraft, gravity), :245-259 (hub), :291-298
 * (disturbance torque), :301-310 + actuatorPrimatives.py:7-63 (wheels).  */
#define NX (12 + BSK_MAX_RW)

static void gravity(orc_ctx* ctx, const double r[3], double t, const double sun[3], double a[3]) {
    const bsk_config* c = ctx->c;
    if (c->gravity_model == BSK_GRAV_SH) {
        /* planet-fixed frame = R3(planet_rate * t) from inertial */
        double th = c->planet_rate * t, ct = cos(th), st = sin(th);
        double p[3] = {ct * r[0] + st * r[1], -st * r[0] + ct * r[1], r[2]}, ap[3];
        sh_field(ctx, p, ap);
        a[0] = ct * ap[0] - st * ap[1]; a[1] = st * ap[0] + ct * ap[1]; a[2] = ap[2];
    } else {
        double rm = v3norm(r), r3 = rm * rm * rm;
        v3scale(-c->mu / r3, r, a);
        if (c->gravity_model == BSK_GRAV_PM_J2) {
            /* closed-form J2 */
            double r5 = r3 * rm * rm, z2 = r[2] * r[2] / (rm * rm);
            double k = 1.5 * c->j2 * c->mu * c->req * c->req / r5;
            a[0] += k * r[0] * (5.0 * z2 - 1.0);
            a[1] += k * r[1] * (5.0 * z2 - 1.0);
            a[2] += k * r[2] * (5.0 * z2 - 3.0);
        }
    }
    if (c->flags & BSK_FLAG_SUN_THIRD_BODY) {
        /* third-body perturbation relative to the central body (…Simulator.py:227-229) */
        double d[3]; v3sub(sun, r, d);
        double dm = v3norm(d), sm = v3norm(sun);
        for (int k = 0; k < 3; ++k) a[k] += c->mu_sun * (d[k] / (dm * dm * dm) - sun[k] / (sm * sm * sm));
    }
}

/* Wheel torque acting along each spin axis = held motor torque + Coulomb friction.  Like the
 * motor command, the friction torque is evaluated once per effector update (dyn tick) from the
 * wheel speed at that time and held through the integrator's stages (Basilisk computes both in
 * the RW effector's ConfigureRWRequests, outside the equations of motion) [BSK-recall]. */
static void wheel_torque(orc_ctx* ctx, const double x[NX], const double u[BSK_MAX_RW], double tq[BSK_MAX_RW]) {
    const bsk_config* c = ctx->c;
    for (int i = 0; i < BSK_MAX_RW; ++i) tq[i] = 0.0;
    for (int i = 0; i < c->n_rw; ++i) {
        double Om = x[12 + i], fr = 0.0;
        if (Om > 0.0) fr = -c->f_coulomb; else if (Om < 0.0) fr = c->f_coulomb;
        tq[i] = u[i] + fr;
    }
}

/* facetDragDynamicEffector (…Simulator.py:272-284): per facet, projected area A n.v_hat (only when
 * positive), force -1/2 rho |v|^2 Cd A_proj v_hat in the body frame, torque r_facet x F; v is the
 * inertial velocity expressed in the body frame.  rho comes from exponentialAtmosphere
 * (…Simulator.py:265-270, parameters :146-148), updated once per dyn tick. */
static void facet_drag(const bsk_config* c, const double sigma[3], const double v_N[3], double rho, double F_B[3], double L_B[3]) {
    double bn[9], vB[3];
    v3set(0, 0, 0, F_B); v3set(0, 0, 0, L_B);
    mrp2c(sigma, bn);
    m33v3(bn, v_N, vB);
    double vm = v3norm(vB);
    if (!(vm > 0.0)) return;
    double vhat[3]; v3scale(1.0 / vm, vB, vhat);
    for (int i = 0; i < c->n_facets; ++i) {
        double proj = c->facet_area[i] * v3dot(c->facet_normal[i], vhat);
        if (proj > 0.0) {
            double f[3], tq[3];
            v3scale(-0.5 * vm * vm * c->facet_cd[i] * proj * rho, vhat, f);
            v3cross(c->facet_pos[i], f, tq);
            v3add(F_B, f, F_B); v3add(L_B, tq, L_B);
        }
    }
}

/* thrusterDynamicEffector (ideal thrusters, …Simulator.py:313-318): thruster i delivers its full
 * thrust while the integrator's time lies within [burst start, burst start + on-time]; time is
 * counted in half dyn steps so that the test is exact: e2 = 2 (tick - tick0) + {0,1,1,2}. */
typedef struct { int active; int e2; const double* lim; } thr_state;

static void eom(orc_ctx* ctx, const double x[NX], const double tq[BSK_MAX_RW], const double lext_in[3], double t,
                const double sun[3], double rho, const thr_state* th, double dx[NX]) {
    const bsk_config* c = ctx->c;
    const double *r = x, *v = x + 3, *sg = x + 6, *w = x + 9, *Om = x + 12;
    double lext[3];
    v3copy(lext_in, lext);
    /* translation */
    v3copy(v, dx);
    gravity(ctx, r, t, sun, dx + 3);
    if (c->flags & BSK_FLAG_DRAG) {
        double F_B[3], L_B[3], bn[9], F_N[3];
        facet_drag(c, sg, v, rho, F_B, L_B);
        mrp2c(sg, bn);
        for (int i = 0; i < 3; ++i) F_N[i] = bn[i] * F_B[0] + bn[3 + i] * F_B[1] + bn[6 + i] * F_B[2];
        for (int i = 0; i < 3; ++i) dx[3 + i] += F_N[i] / c->mass;
        v3add(lext, L_B, lext);
    }
    if (th && th->active) {
        double F_B[3] = {0, 0, 0}, bn[9];
        for (int i = 0; i < c->n_thr; ++i)
            if (th->lim[i] > 0.0 && (double)th->e2 <= th->lim[i]) {
                v3add(F_B, ctx->thr_f[i], F_B);
                v3add(lext, ctx->thr_l[i], lext);
            }
        mrp2c(sg, bn);
        for (int i = 0; i < 3; ++i) dx[3 + i] += (bn[i] * F_B[0] + bn[3 + i] * F_B[1] + bn[6 + i] * F_B[2]) / c->mass;
    }
    /* MRP kinematics: sigma' = 1/4 [(1 - s^2) I + 2 s~ + 2 s s^T] omega */
    double s2 = v3dot(sg, sg), sw = v3dot(sg, w), cx[3];
    v3cross(sg, w, cx);
    for (int k = 0; k < 3; ++k) dx[6 + k] = 0.25 * ((1.0 - s2) * w[k] + 2.0 * cx[k] + 2.0 * sw * sg[k]);
    /* rotation, balanced wheels: back-substitution
       [I - sum Js g g^T] w' = -w x (I w) - sum [ g tq + Js Omega (w x g) ] + L_ext              */
    double Iw[3], rhs[3];
    m33v3(c->inertia, w, Iw);
    v3cross(w, Iw, rhs);
    v3scale(-1.0, rhs, rhs);
    v3add(rhs, lext, rhs);
    for (int i = 0; i < c->n_rw; ++i) {
        double wg[3]; v3cross(w, c->gs[i], wg);
        for (int k = 0; k < 3; ++k) rhs[k] -= c->gs[i][k] * tq[i] + c->js[i] * Om[i] * wg[k];
    }
    m33v3(ctx->dinv, rhs, dx + 9);
    for (int i = 0; i < c->n_rw; ++i) dx[12 + i] = tq[i] / c->js[i] - v3dot(c->gs[i], dx + 9);
    for (int i = c->n_rw; i < BSK_MAX_RW; ++i) dx[12 + i] = 0.0;
}

/* classic RK4 (Basilisk default integrator svIntegratorRK4; the reference never selects another,
 * …Simulator.py:213-214), then the MRP shadow-set switch once per completed step. */
static void rk4_step(orc_ctx* ctx, double x[NX], const double ucmd[BSK_MAX_RW], const double lext[3], double t, double h,
                     const double sun[3], thr_state* th) {
    double k[NX], xt[NX], acc[NX], u[BSK_MAX_RW];
    /* exponentialAtmosphere: density at the spacecraft's position, refreshed once per dyn tick */
    double rho = 0.0;
    if (ctx->c->flags & BSK_FLAG_DRAG) rho = ctx->c->base_density * exp(-(v3norm(x) - ctx->c->req) / ctx->c->scale_height);
    wheel_torque(ctx, x, ucmd, u); /* motor + friction torque, held over the step */
    const int e2 = th ? th->e2 : 0;
    eom(ctx, x, u, lext, t, sun, rho, th, k);
    for (int i = 0; i < NX; ++i) { acc[i] = x[i] + h / 6.0 * k[i]; xt[i] = x[i] + 0.5 * h * k[i]; }
    if (th) th->e2 = e2 + 1;
    if (g_friction_per_stage) wheel_torque(ctx, xt, ucmd, u);
    eom(ctx, xt, u, lext, t + 0.5 * h, sun, rho, th, k);
    for (int i = 0; i < NX; ++i) { acc[i] += h / 3.0 * k[i]; xt[i] = x[i] + 0.5 * h * k[i]; }
    if (g_friction_per_stage) wheel_torque(ctx, xt, ucmd, u);
    eom(ctx, xt, u, lext, t + 0.5 * h, sun, rho, th, k);
    for (int i = 0; i < NX; ++i) { acc[i] += h / 3.0 * k[i]; xt[i] = x[i] + h * k[i]; }
    if (th) th->e2 = e2 + 2;
    if (g_friction_per_stage) wheel_torque(ctx, xt, ucmd, u);
    eom(ctx, xt, u, lext, t + h, sun, rho, th, k);
    for (int i = 0; i < NX; ++i) x[i] = acc[i] + h / 6.0 * k[i];
    double s2 = v3dot(x + 6, x + 6);
    if (s2 > 1.0) v3scale(-1.0 / s2, x + 6, x + 6);
}

/* ---------------------------------------------------------------- FSW chain (1 Hz)
 * hillPoint | inertial3D -> attTrackingError -> MRP_Feedback -> rwMotorTorque, as wired at
 * reference leoPowerAttitudeSimulator.py:407-449 with the mode logic of :548-588. */
typedef struct { double sigma_BR[3], omega_BR_B[3], omega_RN_B[3], domega_RN_B[3]; } att_guid;

static void guidance(orc_ctx* ctx, const double x[NX], int action, att_guid* g) {
    const bsk_config* c = ctx->c;
    double sigma_RN[3], omega_RN_N[3], domega_RN_N[3];
    if (action == 0 && v3norm(x) == 0.0) {
        /* hillPoint on a navigation message nobody has written yet (all zeros; the FSW tasks' first tick, see
           orc_step): its unit vectors normalise to zero, the DCM of zeros maps to the zero MRP and the module's
           radius guard zeroes the rates [BSK-recall] */
        v3set(0, 0, 0, sigma_RN); v3set(0, 0, 0, omega_RN_N); v3set(0, 0, 0, domega_RN_N);
    } else if (action == 0) {
        /* hillPoint (…Simulator.py:414-419): Hill frame {i_r, i_theta, i_h} */
        const double *r = x, *v = x + 3;
        double rm = v3norm(r), h[3], dcm[9];
        v3cross(r, v, h);
        double hm = v3norm(h);
        v3scale(1.0 / rm, r, dcm);
        v3scale(1.0 / hm, h, dcm + 6);
        v3cross(dcm + 6, dcm, dcm + 3);
        c2mrp(dcm, sigma_RN);
        double dfdt = hm / (rm * rm), ddfdt2 = -2.0 * v3dot(v, dcm) / rm * dfdt;
        v3scale(dfdt, dcm + 6, omega_RN_N);
        v3scale(ddfdt2, dcm + 6, domega_RN_N);
    } else {
        /* inertial3D (…Simulator.py:407-411, sigma_R0N :170) — also the attitude target of the
           desat mode (action 2, :574-588) */
        v3copy(c->sigma_R0N, sigma_RN);
        v3set(0, 0, 0, omega_RN_N); v3set(0, 0, 0, domega_RN_N);
    }
    /* attTrackingError (…Simulator.py:422-428) */
    double bn[9];
    submrp(x + 6, sigma_RN, g->sigma_BR);
    mrp2c(x + 6, bn);
    m33v3(bn, omega_RN_N, g->omega_RN_B);
    m33v3(bn, domega_RN_N, g->domega_RN_B);
    v3sub(x + 9, g->omega_RN_B, g->omega_BR_B);
}

static void control(orc_ctx* ctx, const att_guid* g, double u[BSK_MAX_RW]) {
    const bsk_config* c = ctx->c;
    /* MRP_Feedback (…Simulator.py:440-449; K,P :178-180; Ki<0 -> no integral; no wheel-speed
       input is wired in this scenario, so the gyroscopic term uses I*omega only) */
    double w_BN[3], Lr[3], t1[3], t2[3], t3[3];
    v3add(g->omega_BR_B, g->omega_RN_B, w_BN);
    for (int k = 0; k < 3; ++k) Lr[k] = c->K * g->sigma_BR[k] + c->P * g->omega_BR_B[k];
    m33v3(c->inertia, w_BN, t1);
    v3cross(g->omega_RN_B, t1, t2);
    v3sub(Lr, t2, Lr);
    v3cross(w_BN, g->omega_RN_B, t1);
    v3sub(t1, g->domega_RN_B, t2);
    m33v3(c->inertia, t2, t3);
    v3add(Lr, t3, Lr);
    v3scale(-1.0, Lr, Lr); /* torque to apply on the body */
    /* rwMotorTorque (…Simulator.py:431-437): u_s = -[CGs]^T([CGs][CGs]^T)^-1 [C] Lr, then the
       wheel's own saturation and dead-band (reactionWheelStateEffector, HR16 preset) */
    for (int i = 0; i < c->n_rw; ++i) {
        double us = -v3dot(ctx->map[i], Lr);
        if (c->u_max > 0.0) { if (us > c->u_max) us = c->u_max; else if (us < -c->u_max) us = -c->u_max; }
        if (fabs(us) < c->u_min) us = 0.0;
        u[i] = us;
    }
    for (int i = c->n_rw; i < BSK_MAX_RW; ++i) u[i] = 0.0;
}

/* ---------------------------------------------------------------- context set-up */
static int ctx_init(orc_ctx* ctx, const bsk_config* c, const double* cbar, const double* sbar) {
    memset(ctx, 0, sizeof(*ctx));
    ctx->c = c;
    double d[9];
    memcpy(d, c->inertia, sizeof d);
    for (int i = 0; i < c->n_rw; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) d[3 * a + b] -= c->js[i] * c->gs[i][a] * c->gs[i][b];
    if (m33inv(d, ctx->dinv)) return -1;
    if (c->n_rw > 0) {
        /* CGs = C * Gs (3 x n);  map = CGs^T (CGs CGs^T)^-1 C  (n x 3) */
        double cgs[3][BSK_MAX_RW], m[9] = {0}, mi[9];
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < c->n_rw; ++i)
                cgs[a][i] = c->ctrl_axes[3 * a] * c->gs[i][0] + c->ctrl_axes[3 * a + 1] * c->gs[i][1] + c->ctrl_axes[3 * a + 2] * c->gs[i][2];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                for (int i = 0; i < c->n_rw; ++i) m[3 * a + b] += cgs[a][i] * cgs[b][i];
        if (m33inv(m, mi)) return -1;
        for (int i = 0; i < c->n_rw; ++i) {
            double tmp[3];
            for (int a = 0; a < 3; ++a) tmp[a] = cgs[0][i] * mi[a] + cgs[1][i] * mi[3 + a] + cgs[2][i] * mi[6 + a];
            for (int b = 0; b < 3; ++b)
                ctx->map[i][b] = tmp[0] * c->ctrl_axes[b] + tmp[1] * c->ctrl_axes[3 + b] + tmp[2] * c->ctrl_axes[6 + b];
        }
    }
    if (c->flags & BSK_FLAG_DESAT) {
        double dd[9] = {0}, ddi[9], D[BSK_MAX_THR][3];
        for (int i = 0; i < c->n_thr; ++i) {
            v3cross(c->thr_pos[i], c->thr_dir[i], D[i]);
            v3scale(c->thr_max_thrust, c->thr_dir[i], ctx->thr_f[i]);
            v3scale(c->thr_max_thrust, D[i], ctx->thr_l[i]);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) dd[3 * a + b] += D[i][a] * D[i][b];
        }
        if (m33inv(dd, ddi)) return -1;
        for (int i = 0; i < c->n_thr; ++i) m33v3(ddi, D[i], ctx->thr_map[i]);   /* (DD^T)^-1 is symmetric */
    }
    if (c->gravity_model == BSK_GRAV_SH) {
        if (!cbar || !sbar || c->sh_degree < 2 || c->sh_degree > BSK_MAX_SH_DEGREE) return -1;
        ctx->deg = c->sh_degree; ctx->cbar = cbar; ctx->sbar = sbar;
        sh_init(ctx);
    }
    return 0;
}
static void ctx_free(orc_ctx* ctx) { free(ctx->abar); free(ctx->n1); free(ctx->n2); free(ctx->nq1); free(ctx->nq2); }

/* ---------------------------------------------------------------- power system ("next" row f1)
 * eclipse (conical Earth shadow) -> simpleSolarPanel -> simpleBattery <- simplePowerSink
 * (…Simulator.py:286-288, 326-345; parameters :158-167), Euler-integrated at the dyn rate. */
static double safe_asin(double x) { return x >= 1.0 ? M_PI / 2 : (x <= -1.0 ? -M_PI / 2 : asin(x)); }

/* fraction of the solar disc left visible, from the apparent radii a (Sun), b (planet) and the
   apparent separation c of their centres as seen from the spacecraft: the published two-disc lens area
       x = (c^2 + a^2 - b^2)/(2c),  y = sqrt(a^2 - x^2),  area = a^2 acos(x/a) + b^2 acos((c - x)/b) - c y.
   With the planet's disc a thousand times the Sun's, (c - x)/b sits within 1e-8 of 1 and acos() amplifies its
   rounding by 1e8: evaluated as written the factor is off by 1e-9 typically and 6e-8 near first contact in fp64
   (seed 7777 of tests/test_gpu_fuzz.py), and still by 1e-10 in x87 extended precision (seed 96 of
   tests/test_gpu_fuzz_wide.py) - in both cases the 50-digit value sided with the kernel.  So the two angles are
   taken from the chord's half-height y instead - acos(x/a) = atan2(y, x), acos((c - x)/b) = atan2(y, c - x): the
   same lens, without the amplification - and in long double.  Basilisk's own fp64 evaluation of the written form
   carries the noise; that is not something to reproduce. */
static double percent_shadow(double req, const double r_HB[3], const double s_BP[3]) {
    const long double REQ_SUN = 695000.0e3L, PI_L = 3.14159265358979323846264338327950288L;
    long double nh = sqrtl((long double)r_HB[0] * r_HB[0] + (long double)r_HB[1] * r_HB[1] + (long double)r_HB[2] * r_HB[2]);
    long double ns = sqrtl((long double)s_BP[0] * s_BP[0] + (long double)s_BP[1] * s_BP[1] + (long double)s_BP[2] * s_BP[2]);
    long double sa = REQ_SUN / nh, sb = (long double)req / ns;
    long double a = sa >= 1.0L ? PI_L / 2 : asinl(sa), b = sb >= 1.0L ? PI_L / 2 : asinl(sb);
    /* separation of the two directions through the cross product: acos() of their cosine loses digits when the
       spacecraft looks almost along the Sun line */
    long double u[3] = {-(long double)s_BP[0], -(long double)s_BP[1], -(long double)s_BP[2]};
    long double w[3] = {r_HB[0], r_HB[1], r_HB[2]};
    long double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
    long double c = atan2l(sqrtl(cx * cx + cy * cy + cz * cz), u[0] * w[0] + u[1] * w[1] + u[2] * w[2]);
    if (c < b - a) return 0.0;                                   /* total */
    if (c < a - b) return (double)(1.0L - (b * b) / (a * a));    /* annular */
    if (c < a + b) {                                             /* partial: lens area of two discs */
        long double x = (c * c + a * a - b * b) / (2.0L * c), y2 = a * a - x * x, y = y2 > 0.0L ? sqrtl(y2) : 0.0L;
        long double area = a * a * atan2l(y, x) + b * b * atan2l(y, c - x) - c * y;
        return (double)(1.0L - area / (PI_L * a * a));
    }
    return 1.0;
}

/* The same fraction exactly as the eclipse module writes it [BSK-recall: computePercentShadow], in plain fp64:
   a = safeAsin(R_sun/|r_HB|), b = safeAsin(R_p/|s_BP|), c = safeAcos(-s_BP.r_HB/(|s_BP||r_HB|)),
   area = a^2 acos(x/a) + b^2 acos((c - x)/b) - c y.  Selected with orc_set_penumbra_form(1): what the reference engine
   itself would report on a penumbra tick, ill-conditioning included (tests/test_oracle_power.py records how far that
   sits from the conditioned form above). */
static double safe_acos(double x) { return x >= 1.0 ? 0.0 : (x <= -1.0 ? M_PI : acos(x)); }
static double percent_shadow_as_written(double req, const double r_HB[3], const double s_BP[3]) {
    const double REQ_SUN = 695000.0e3;
    double shadowFraction = 1.0;
    double normR_HB = v3norm(r_HB), normS_BP = v3norm(s_BP);
    double a = safe_asin(REQ_SUN / normR_HB);
    double b = safe_asin(req / normS_BP);
    double c = safe_acos(-v3dot(s_BP, r_HB) / (normS_BP * normR_HB));
    if (c < b - a) {
        shadowFraction = 0.0;
    } else if (c < a - b) {
        double areaSun = M_PI * a * a, areaBody = M_PI * b * b;
        shadowFraction = 1 - (areaSun - areaBody) / (M_PI * a * a);
    } else if (c < a + b) {
        double x = (c * c + a * a - b * b) / (2 * c);
        double y = sqrt(a * a - x * x);
        double area = a * a * safe_acos(x / a) + b * b * safe_acos((c - x) / b) - c * y;
        shadowFraction = 1 - area / (M_PI * a * a);
    }
    return shadowFraction;
}
static int g_penumbra_form = 0;   /* 0: conditioned form (default, what the kernels are held to); 1: as written, fp64 */

static double shadow_factor(const bsk_config* c, const double r[3], const double sun[3]) {
    /* Earth is the zero base (…Simulator.py:225): s_BP = r, r_HP = sun, r_HB = sun - r */
    const double REQ_SUN = 695000.0e3;
    double r_HB[3]; v3sub(sun, r, r_HB);
    double nhp = v3norm(sun);
    if (v3norm(r_HB) < nhp) return 1.0;              /* spacecraft on the day side of the planet */
    double f1 = safe_asin((REQ_SUN + c->req) / nhp), f2 = safe_asin((REQ_SUN - c->req) / nhp);
    double s = v3norm(r), s0 = -v3dot(r, sun) / nhp;
    double c1 = s0 + c->req / sin(f1), c2 = s0 - c->req / sin(f2);
    double l = sqrt(s * s - s0 * s0), l1 = c1 * tan(f1), l2 = c2 * tan(f2);
    if (fabs(l) < fabs(l2) || fabs(l) < fabs(l1))
        return g_penumbra_form ? percent_shadow_as_written(c->req, r_HB, r) : percent_shadow(c->req, r_HB, r);
    return 1.0;
}

/* ---------------------------------------------------------------- batched driver
 * One env step = run_sim(action) (…Simulator.py:535-644) + the env's reward / done logic
 * (leoPowerAttitudeEnvironment.py:98-127, 161-170) for each of n spacecraft.
 * state: host SoA [n_fields][n] with the field order of include/bskgpu.h.                    */
int orc_n_fields(const bsk_config* c) { return BSK_NF_BASE + c->n_rw + BSK_NF_TAIL; }

int orc_step(const bsk_config* c, int n, double* state, int32_t* steps, int32_t* ticks, const int32_t* actions,
             int substeps, double sim_time0, const double* cbar, const double* sbar, double* obs, double* reward,
             uint8_t* done, uint8_t* reason) {
    orc_ctx ctx;
    if (ctx_init(&ctx, c, cbar, sbar)) return -1;
    const int nrw = c->n_rw, tail = BSK_NF_BASE + nrw;
#define S(f, i) state[(size_t)(f) * n + (i)]
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int e = 0; e < n; ++e) {
        double x[NX] = {0}, u[BSK_MAX_RW], lext[3];
        for (int f = 0; f < 12 + nrw; ++f) x[f] = S(f, e);
        for (int k = 0; k < 3; ++k) lext[k] = S(tail + BSK_T_LEXT + k, e);
        for (int i = 0; i < BSK_MAX_RW; ++i) u[i] = S(tail + BSK_T_UCMD + i, e);
        /* torque the next FSW tick will command (fsw_lag; zero after a reset = empty att_guidance message) */
        double upend[BSK_MAX_RW];
        for (int i = 0; i < BSK_MAX_RW; ++i) upend[i] = S(tail + BSK_T_UPEND + i, e);
        double charge = S(tail + BSK_T_CHARGE, e), shadow = 1.0;
        int act = actions[e], tick = ticks[e];
        /* desaturation state (row f2) */
        const int desat = (c->flags & BSK_FLAG_DESAT) != 0;
        double thr_rem[BSK_MAX_THR], thr_lim[BSK_MAX_THR];
        for (int i = 0; i < BSK_MAX_THR; ++i) { thr_rem[i] = S(tail + BSK_T_THR_REM + i, e); thr_lim[i] = S(tail + BSK_T_THR_LIM + i, e); }
        int thr_t0 = (int)S(tail + BSK_T_THR_T0, e), thr_cnt = (int)S(tail + BSK_T_THR_CNT, e), first_fsw = 1;
        /* Sun position: evaluated at the start of the env step from this spacecraft's own clock and
           held over the step, like the 180 s SPICE task (…Simulator.py:102,357) */
        double sun[3];
        for (int k = 0; k < 3; ++k) sun[k] = (c->sun_r0[k] + c->sun_v[k] * sim_time0) + c->sun_v[k] * (tick * c->dt);
        /* What the dynamics task has latched from the FSW tasks' output messages: wheel torque command u and the
           thruster burst (thr_lim, thr_t0).  The FSW chain itself works on `u_cmd_msg` / `lim_msg`; with nav_lag the
           effectors see them one integrator step later (below). */
        double sbr = S(tail + BSK_T_SBR, e);          /* |sigma_BR| of the att_guidance message */
        double u_msg[BSK_MAX_RW], lim_msg[BSK_MAX_THR];
        int have_msg = 0, lim_changed = 0;
#define FSW_TICK(NAV)                                                                                                    \
        do {                                                                                                             \
            /* mrpControlTask runs MRP_Feedback, attTrackingError, rwMotorTorque in THAT order (AddModelToTask calls,     \
               …Simulator.py:484-486): the controller reads the att_guidance message the previous FSW tick wrote, then  \
               this tick's guidance overwrites it.  MRP_Feedback without integral term and rwMotorTorque are pure        \
               functions of the message, so the slab keeps the 4 torques it maps to rather than its 12 entries. */       \
            att_guid g;                                                                                                  \
            guidance(&ctx, (NAV), act, &g);                                                                              \
            sbr = v3norm(g.sigma_BR);                                                                                    \
            if (c->fsw_lag) {                                                                                            \
                for (int i = 0; i < BSK_MAX_RW; ++i) u_msg[i] = upend[i];                                                \
                control(&ctx, &g, upend);                                                                                \
            } else {                                                                                                     \
                control(&ctx, &g, u_msg);                                                                                \
            }                                                                                                            \
            have_msg = 1;                                                                                                \
            if (desat && act == 2) {                                                                                     \
                /* rwDesatTask (…Simulator.py:452-478, 488-490; enabled in mode 2 only, :574-588) */                     \
                const double Tc = c->fsw_every * c->dt;                                                                  \
                if (first_fsw) {                                                                                         \
                    /* thrMomentumManagement: one request per mode entry (its Reset, :580).  Wheel momentum              \
                       h_s = sum Js Om g from the wheel-speed message; dump everything above hs_min (:183). */           \
                    double hs[3] = {0, 0, 0};                                                                            \
                    for (int i = 0; i < nrw; ++i)                                                                        \
                        for (int k = 0; k < 3; ++k) hs[k] += c->js[i] * (NAV)[12 + i] * c->gs[i][k];                     \
                    double hm = v3norm(hs), dH[3] = {0, 0, 0};                                                           \
                    if (hm > c->hs_min) v3scale(-(hm - c->hs_min) / hm, hs, dH);                                         \
                    /* thrForceMapping, on-pulsing (thrForceSign +1, :186): minimum-norm impulses                        \
                       F = D^T (D D^T)^-1 dH, then subtract the smallest so that all are >= 0 */                         \
                    double F[BSK_MAX_THR], fmin = 0.0;                                                                   \
                    for (int i = 0; i < c->n_thr; ++i) { F[i] = v3dot(ctx.thr_map[i], dH); if (i == 0 || F[i] < fmin) fmin = F[i]; } \
                    /* thrMomentumDumping: a new request resets the schedule (Reset, :581) */                            \
                    for (int i = 0; i < c->n_thr; ++i) thr_rem[i] = (F[i] - fmin) / c->thr_max_thrust;                   \
                    thr_cnt = 0;                                                                                         \
                }                                                                                                        \
                if (thr_cnt <= 0) {                                                                                      \
                    /* fire: each thruster for min(remaining, control period); pulses shorter than thrMinFireTime       \
                       (:190) are dropped; the thruster stretches a pulse to its MinOnTime */                           \
                    for (int i = 0; i < BSK_MAX_THR; ++i) lim_msg[i] = thr_lim[i];                                       \
                    for (int i = 0; i < c->n_thr; ++i) {                                                                 \
                        double on = thr_rem[i] < Tc ? thr_rem[i] : Tc;                                                   \
                        if (on < c->thr_min_fire_time) { on = 0.0; thr_rem[i] = 0.0; lim_msg[i] = 0.0; continue; }       \
                        thr_rem[i] -= on;                                                                                \
                        if (on >= Tc) lim_msg[i] = 2.0 * c->fsw_every;                                                   \
                        else { if (on < c->thr_min_on_time) on = c->thr_min_on_time; lim_msg[i] = floor(on * (2.0 / c->dt)); } \
                    }                                                                                                    \
                    lim_changed = 1;                                                                                     \
                    thr_cnt = c->thr_max_counter;                                                                        \
                } else {                                                                                                 \
                    thr_cnt -= 1;                                                                                        \
                }                                                                                                        \
            }                                                                                                            \
            first_fsw = 0;                                                                                               \
        } while (0)
        /* the dynamics task's effectors read the FSW output messages after the integration to the current tick:
           the new torque / burst act from tick `tick` on */
#define LATCH()                                                                                                          \
        do {                                                                                                             \
            if (have_msg) for (int i = 0; i < BSK_MAX_RW; ++i) u[i] = u_msg[i];                                          \
            if (lim_changed) { for (int i = 0; i < BSK_MAX_THR; ++i) thr_lim[i] = lim_msg[i]; thr_t0 = tick; }           \
            have_msg = 0; lim_changed = 0;                                                                               \
        } while (0)
        /* Task priorities (bsk_config.nav_lag): the reference gives its FSW tasks priorities 100 / 50 (…Simulator.py:
           383-386) and leaves the dynamics tasks at the default (:101-103), and Basilisk runs higher priorities first
           at equal time [BSK-recall]: an FSW tick at time k dt executes BEFORE the dynamics task integrates to that
           time, on the navigation / wheel-speed messages of time (k-1) dt; its commands are latched when the dynamics
           task runs (from time k dt on).  The tick at t = 0 finds messages nobody has written (zeros); a tick that
           coincides with the end of an env step belongs to THAT step (ExecuteSimulation runs the tasks scheduled at
           its stop time).  nav_lag = 0: the FSW tick at time k dt works on the state at k dt and belongs to the env
           step that starts there. */
        const int navlag = c->nav_lag && nrw > 0;
        if (navlag && tick == 0) {
            double nav0[NX] = {0};
            if (g_t0_real_messages) for (int f = 0; f < NX; ++f) nav0[f] = x[f];
            FSW_TICK(nav0);
            LATCH();
        }
        for (int j = 0; j < substeps; ++j) {
            if (nrw > 0 && !navlag && tick % c->fsw_every == 0) { FSW_TICK(x); LATCH(); }
            if (navlag && (tick + 1) % c->fsw_every == 0) FSW_TICK(x);
            double t = tick * c->dt;
            thr_state th = {0, 0, thr_lim};
            if (desat) {
                th.e2 = 2 * (tick - thr_t0);
                for (int i = 0; i < c->n_thr; ++i) if (thr_lim[i] > 0.0 && (double)th.e2 <= thr_lim[i]) th.active = 1;
            }
            if (g_sun_per_tick)
                for (int k = 0; k < 3; ++k) sun[k] = (c->sun_r0[k] + c->sun_v[k] * sim_time0) + c->sun_v[k] * (tick * c->dt);
            rk4_step(&ctx, x, u, lext, t, c->dt, sun, desat ? &th : 0);
            ++tick;
            if (navlag) LATCH();
            if (c->flags & BSK_FLAG_POWER) {
                /* EnvTask at the dyn rate (…Simulator.py:363-366): eclipse -> panel -> battery */
                shadow = shadow_factor(c, x, sun);
                double bn[9], sB[3], sN[3];
                v3sub(sun, x, sN);
                double d = v3norm(sN);
                v3scale(1.0 / d, sN, sN);
                mrp2c(x + 6, bn);
                m33v3(bn, sN, sB);
                double proj = v3dot(c->panel_normal, sB);
                if (proj < 0) proj = 0;
                const double AU = 149597870700.0;
                double flux = c->solar_flux * (AU / d) * (AU / d);
                double p = flux * proj * shadow * c->panel_area * c->panel_efficiency + c->power_draw;
                charge += p * c->dt;
                if (charge > c->storage_capacity) charge = c->storage_capacity;
                if (charge < 0) charge = 0;
            }
        }
        /* observation (…Simulator.py:636-638, leoPowerAttitudeEnvironment.py:107-108) */
        /* obs[0] is the logged att_guidance message (…Simulator.py:605,611): with nav_lag the one the last FSW tick
           wrote; otherwise the tracking error of the end-of-step state under the step's mode */
        att_guid g;
        guidance(&ctx, x, act, &g);
        double o0 = navlag ? sbr : v3norm(g.sigma_BR), o1 = v3norm(x + 9), o2 = 0;
        for (int i = 0; i < nrw; ++i) o2 += x[12 + i] * x[12 + i];
        o2 = sqrt(o2) / c->wheel_limit;
        double o3 = charge / 3600.0 / c->power_max;
        if (c->flags & BSK_FLAG_POWER) shadow = shadow_factor(c, x, sun);
        /* reward and termination (leoPowerAttitudeEnvironment.py:98-127, 161-170) */
        uint8_t why = 0;
        double rw = (act == 0) ? c->reward_mult / (1.0 + o0 * o0) : 0.0;
        if (steps[e] >= c->max_length) why |= BSK_DONE_LENGTH;
        if (o2 > 1.0) { why |= BSK_DONE_WHEELS; rw -= c->failure_penalty; }
        if (o3 == 0.0) { why |= BSK_DONE_BATTERY; rw -= c->failure_penalty; }
        if (v3norm(x) < c->r_min) why |= BSK_DONE_ORBIT;
        if (obs) { obs[0 * (size_t)n + e] = o0; obs[1 * (size_t)n + e] = o1; obs[2 * (size_t)n + e] = o2;
                   obs[3 * (size_t)n + e] = o3; obs[4 * (size_t)n + e] = shadow; }
        if (reward) reward[e] = rw;
        if (done) done[e] = why != 0;
        if (reason) reason[e] = why;
        for (int f = 0; f < 12 + nrw; ++f) S(f, e) = x[f];
        for (int i = 0; i < BSK_MAX_RW; ++i) S(tail + BSK_T_UCMD + i, e) = u[i];
        for (int i = 0; i < BSK_MAX_RW; ++i) S(tail + BSK_T_UPEND + i, e) = upend[i];
        S(tail + BSK_T_SBR, e) = sbr;
        S(tail + BSK_T_CHARGE, e) = charge;
        if (desat) {
            for (int i = 0; i < BSK_MAX_THR; ++i) { S(tail + BSK_T_THR_REM + i, e) = thr_rem[i]; S(tail + BSK_T_THR_LIM + i, e) = thr_lim[i]; }
            S(tail + BSK_T_THR_T0, e) = (double)thr_t0;
            S(tail + BSK_T_THR_CNT, e) = (double)thr_cnt;
        }
        steps[e] += 1;
        ticks[e] = tick;
    }
#undef S
#undef FSW_TICK
#undef LATCH
    ctx_free(&ctx);
    return 0;
}

#ifdef ORC_OMP
#include <omp.h>
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_get_threads(void) { return omp_get_max_threads(); }
#else
void orc_set_threads(int n) { (void)n; }
int orc_get_threads(void) { return 1; }
#endif

/* Single-function probes used by the known-answer tests. */
int orc_gravity(const bsk_config* c, const double* cbar, const double* sbar, const double r[3], double t, double a[3]) {
    orc_ctx ctx;
    if (ctx_init(&ctx, c, cbar, sbar)) return -1;
    double sun[3];
    for (int k = 0; k < 3; ++k) sun[k] = c->sun_r0[k] + c->sun_v[k] * t;
    gravity(&ctx, r, t, sun, a);
    ctx_free(&ctx);
    return 0;
}
int orc_eom(const bsk_config* c, const double* x, const double* u, const double* lext, double t, double* dx) {
    orc_ctx ctx;
    if (ctx_init(&ctx, c, 0, 0)) return -1;
    double xx[NX] = {0}, uu[BSK_MAX_RW] = {0}, dd[NX];
    memcpy(xx, x, sizeof(double) * (12 + c->n_rw));
    memcpy(uu, u, sizeof(double) * c->n_rw);
    double tq[BSK_MAX_RW];
    wheel_torque(&ctx, xx, uu, tq);
    double sun[3];
    for (int k = 0; k < 3; ++k) sun[k] = c->sun_r0[k] + c->sun_v[k] * t;
    double rho = (c->flags & BSK_FLAG_DRAG) ? c->base_density * exp(-(v3norm(xx) - c->req) / c->scale_height) : 0.0;
    eom(&ctx, xx, tq, lext, t, sun, rho, 0, dd);
    memcpy(dx, dd, sizeof(double) * (12 + c->n_rw));
    ctx_free(&ctx);
    return 0;
}
int orc_fsw(const bsk_config* c, const double* x, int action, double* guid12, double* u) {
    orc_ctx ctx;
    if (ctx_init(&ctx, c, 0, 0)) return -1;
    double xx[NX] = {0}, uu[BSK_MAX_RW];
    memcpy(xx, x, sizeof(double) * (12 + c->n_rw));
    att_guid g;
    guidance(&ctx, xx, action, &g);
    control(&ctx, &g, uu);
    memcpy(guid12, &g, sizeof g);
    memcpy(u, uu, sizeof(double) * c->n_rw);
    ctx_free(&ctx);
    return 0;
}
void orc_mrp2c(const double q[3], double c[9]) { mrp2c(q, c); }
void orc_c2mrp(const double c[9], double q[3]) { c2mrp(c, q); }
void orc_submrp(const double a[3], const double b[3], double q[3]) { submrp(a, b, q); }
double orc_shadow(const bsk_config* c, const double r[3], const double sun[3]) { return shadow_factor(c, r, sun); }
/* penumbra expression used by shadow_factor (and so by orc_step): 0 conditioned (default), 1 as Basilisk writes it */
void orc_set_penumbra_form(int form) { g_penumbra_form = form ? 1 : 0; }
int orc_get_penumbra_form(void) { return g_penumbra_form; }
/* decision switches: 0 friction_per_stage, 1 sun_per_tick, 2 t0_real_messages; value 0 / 1 */
int orc_set_decision(int which, int value) {
    int* g = which == 0 ? &g_friction_per_stage : which == 1 ? &g_sun_per_tick : which == 2 ? &g_t0_real_messages : 0;
    if (!g) return -1;
    *g = value ? 1 : 0;
    return 0;
}
