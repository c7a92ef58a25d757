// bsk_capi.hip — C-ABI of libbskgpu.so (see include/bskgpu.h for the contract and the reference
// interfaces each entry point replaces).  Host side only: handle management, HBM allocation,
// uploads/downloads, launch geometry.  No CPU compute path exists here by design: without a
// gfx950 device bsk_create fails with BSK_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bskgpu.h"
#include "bsk_aux.hpp"
#include "bsk_launch.hpp"
#include "bsk_rollout.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? BSK_ENOMEM : BSK_EHIP,                              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

// how many copies / stream synchronisations this library has issued (bsk_debug_counters: tests assert that the
// device-resident entry points issue none)
std::atomic<long long> g_n_copies{0}, g_n_syncs{0};
#define HIP_COPY(expr) do { g_n_copies.fetch_add(1, std::memory_order_relaxed); HIP_TRY(expr); } while (0)
#define HIP_SYNC(expr) do { g_n_syncs.fetch_add(1, std::memory_order_relaxed); HIP_TRY(expr); } while (0)

bool inv3(const double* m, double* o) {
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(std::fabs(det) > 0.0)) return false;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// Low-precision solar position (Astronomical Almanac), equatorial frame, metres, Earth-centred.
// Stands in for the SPICE de430 lookup at reference leoPowerAttitudeSimulator.py:219-225.
void sun_position(double jd, double out[3]) {
    const double D2R = M_PI / 180.0, AU = 149597870700.0;
    double n = jd - 2451545.0;
    double L = std::fmod(280.460 + 0.9856474 * n, 360.0), g = std::fmod(357.528 + 0.9856003 * n, 360.0) * D2R;
    double lam = (L + 1.915 * std::sin(g) + 0.020 * std::sin(2 * g)) * D2R;
    double eps = (23.439 - 0.0000004 * n) * D2R;
    double R = (1.00014 - 0.01671 * std::cos(g) - 0.00014 * std::cos(2 * g)) * AU;
    out[0] = R * std::cos(lam);
    out[1] = R * std::cos(eps) * std::sin(lam);
    out[2] = R * std::sin(eps) * std::sin(lam);
}

// Fused Pines coefficient stream for gravity_sh (bsk_device.hpp), iteration order
// M = 1..d+1, L = M..d+1, 8 doubles per step:
//   [0] L == M: A[M][M] (diagonal constant);  L == M+1: A[M+1][M]/(u A[M][M]);  else n1[L][M]
//   [1] n2[L][M] (L >= M+2)                      -- recursion A[L][M] = u n1 A[L-1][M] - n2 A[L-2][M]
//   [2,3] M (Cbar, Sbar)[L][M]                   -- a1 / a2 sums            (L <= d)
//   [4,5] nq1[L][M-1] (Cbar, Sbar)[L][M-1]       -- a3 sum                  (L <= d)
//   [6,7] nq2[L-1][M-1] (Cbar, Sbar)[L-1][M-1]   -- a4 sum                  (L >= 2)
// Constants as Basilisk's gravityEffector documents them (SURVEY.md §8 note N1).
void build_sh_table(int d, const double* cbar, const double* sbar, std::vector<double>& tab) {
    auto K = [](int i) { return i == 0 ? 1.0 : 2.0; };
    auto idx = [](int l, int m) { return l * (l + 1) / 2 + m; };
    std::vector<double> diag(d + 2), sd(d + 2);
    diag[0] = 1.0;
    for (int l = 1; l <= d + 1; ++l) diag[l] = std::sqrt((double)(2 * l + 1) * K(l) / ((double)(2 * l) * K(l - 1))) * diag[l - 1];
    for (int l = 1; l <= d + 1; ++l) sd[l] = std::sqrt((double)(2 * l) * K(l - 1) / K(l)) * diag[l];
    auto n1 = [](int l, int m) { return std::sqrt((double)(2 * l + 1) * (double)(2 * l - 1) / ((double)(l - m) * (double)(l + m))); };
    auto n2 = [](int l, int m) {
        return std::sqrt((double)(l + m - 1) * (double)(2 * l + 1) * (double)(l - m - 1) /
                         ((double)(l + m) * (double)(l - m) * (double)(2 * l - 3)));
    };
    auto nq1 = [&](int l, int m) { return std::sqrt((double)(l - m) * K(m) * (double)(l + m + 1) / K(m + 1)); };
    auto nq2 = [&](int l, int m) {
        return std::sqrt((double)(l + m + 2) * (double)(l + m + 1) * (double)(2 * l + 1) * K(m) / ((double)(2 * l + 3) * K(m + 1)));
    };
    tab.clear();
    tab.reserve((size_t)(d + 1) * (d + 2) / 2 * 8);
    for (int M = 1; M <= d + 1; ++M)
        for (int L = M; L <= d + 1; ++L) {
            double e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (L == M) e[0] = diag[M];
            else if (L == M + 1) e[0] = sd[M + 1] / diag[M];
            else { e[0] = n1(L, M); e[1] = n2(L, M); }
            if (L <= d) {
                e[2] = M * cbar[idx(L, M)];
                e[3] = M * sbar[idx(L, M)];
                const double q = nq1(L, M - 1);
                e[4] = q * cbar[idx(L, M - 1)];
                e[5] = q * sbar[idx(L, M - 1)];
            }
            if (L >= 2) {
                const double q = nq2(L - 1, M - 1);
                e[6] = q * cbar[idx(L - 1, M - 1)];
                e[7] = q * sbar[idx(L - 1, M - 1)];
            }
            tab.insert(tab.end(), e, e + 8);
        }
    tab.insert(tab.end(), 16, 0.0);   // spare entries: the kernel's software pipeline reads ahead
}

// Stream of the DPP-broadcast form (bsk_device.hpp: gravity_sh_dpp): same iteration order, 8 doubles
// per entry, but (i) the recursion is rescaled column by column, Bt_L = B_L / alpha_L with
// alpha_M = alpha_(M+1) = 1, alpha_L = n2(L, M) alpha_(L-2), so entry[0] = n1 alpha_(L-1) / alpha_L is
// the only recursion constant and the six coefficient products carry alpha_L; (ii) every column is
// padded to an even number of entries (a 128-byte chunk = 2 entries never straddles a column);
// (iii) the stream is padded to whole SH_RING-chunk bodies plus two bodies of read-ahead slack.
// The walk is cut into two halves of (nearly) equal entry count at a column boundary: `split` is the first
// column of the second half, `chunk1` its first chunk; the kernels add the halves' partial sums in a fixed
// order whether one wave or two walk them.
struct ShLayout {
    int split, chunk1, bodies, bodies0, bodies1;
};
ShLayout build_sh_table_dpp(int d, const double* cbar, const double* sbar, std::vector<double>& tab) {
    auto K = [](int i) { return i == 0 ? 1.0L : 2.0L; };
    auto idx = [](int l, int m) { return l * (l + 1) / 2 + m; };
    std::vector<long double> diag(d + 2), sd(d + 2), alpha(d + 3);
    diag[0] = 1.0L;
    for (int l = 1; l <= d + 1; ++l) diag[l] = sqrtl((long double)(2 * l + 1) * K(l) / ((long double)(2 * l) * K(l - 1))) * diag[l - 1];
    for (int l = 1; l <= d + 1; ++l) sd[l] = sqrtl((long double)(2 * l) * K(l - 1) / K(l)) * diag[l];
    auto n1 = [](int l, int m) { return sqrtl((long double)(2 * l + 1) * (long double)(2 * l - 1) / ((long double)(l - m) * (long double)(l + m))); };
    auto n2 = [](int l, int m) {
        return sqrtl((long double)(l + m - 1) * (long double)(2 * l + 1) * (long double)(l - m - 1) /
                     ((long double)(l + m) * (long double)(l - m) * (long double)(2 * l - 3)));
    };
    auto nq1 = [&](int l, int m) { return sqrtl((long double)(l - m) * K(m) * (long double)(l + m + 1) / K(m + 1)); };
    auto nq2 = [&](int l, int m) {
        return sqrtl((long double)(l + m + 2) * (long double)(l + m + 1) * (long double)(2 * l + 1) * K(m) /
                     ((long double)(2 * l + 3) * K(m + 1)));
    };
    tab.clear();
    std::vector<size_t> col_chunk(d + 3, 0);   // first chunk of column M
    for (int M = 1; M <= d + 1; ++M) {
        col_chunk[M] = tab.size() / 16;
        for (int L = M; L <= d + 1; ++L) {
            long double e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            alpha[L] = (L <= M + 1) ? 1.0L : n2(L, M) * alpha[L - 2];
            if (L == M) e[0] = diag[M];
            else if (L == M + 1) e[0] = sd[M + 1] / diag[M];
            else e[0] = n1(L, M) * alpha[L - 1] / alpha[L];
            if (L <= d) {
                e[2] = M * (long double)cbar[idx(L, M)];
                e[3] = M * (long double)sbar[idx(L, M)];
                const long double q = nq1(L, M - 1);
                e[4] = q * cbar[idx(L, M - 1)];
                e[5] = q * sbar[idx(L, M - 1)];
            }
            if (L >= 2) {
                const long double q = nq2(L - 1, M - 1);
                e[6] = q * cbar[idx(L - 1, M - 1)];
                e[7] = q * sbar[idx(L - 1, M - 1)];
            }
            for (int k = 0; k < 8; ++k) tab.push_back((double)(k >= 2 ? e[k] * alpha[L] : e[k]));
        }
        if ((d + 1 - M + 1) & 1) tab.insert(tab.end(), 8, 0.0);   // odd column: one all-zero entry
    }
    const size_t chunks = tab.size() / 16, R = bsk::SH_RING;
    col_chunk[d + 2] = chunks;
    ShLayout lay;
    // Balance the halves by issue slots, not by chunks: a column end costs about two chunks' worth (flush,
    // combine, restart, two taken branches) and the second half has many short columns; it also raises
    // (s + i t) to its first column's power first (about a third of a chunk per column skipped).
    auto cost0 = [&](int sp) { return (double)col_chunk[sp] + 2.0 * (sp - 1); };
    auto cost1 = [&](int sp) { return (double)(chunks - col_chunk[sp]) + 2.0 * (d + 2 - sp) + 0.33 * (sp - 1); };
    lay.split = 2;                                   // 1 < split <= d + 1: both halves own at least one column
    while (lay.split < d + 1 && cost0(lay.split + 1) <= cost1(lay.split + 1)) ++lay.split;
    lay.chunk1 = (int)col_chunk[lay.split];
    lay.bodies = (int)((chunks + R - 1) / R);
    lay.bodies0 = (int)((col_chunk[lay.split] + R - 1) / R);
    lay.bodies1 = (int)((chunks - col_chunk[lay.split] + R - 1) / R);
    tab.resize(((size_t)lay.bodies * R + 2 * R) * 16, 0.0);
    return lay;
}

int build_params(const bsk_config& c, bsk::StepParams& p, bsk::ColdCfg& k, bool& diag) {
    std::memset(&p, 0, sizeof p);
    std::memset(&k, 0, sizeof k);
    p.dt = c.dt;
    p.mu = c.mu;
    p.j2k = 1.5 * c.j2 * c.mu * c.req * c.req;
    std::memcpy(p.inertia, c.inertia, sizeof p.inertia);
    std::memcpy(k.inertia, c.inertia, sizeof k.inertia);
    double D[9];
    std::memcpy(D, c.inertia, sizeof D);
    for (int i = 0; i < c.n_rw; ++i) {
        double nrm = std::sqrt(c.gs[i][0] * c.gs[i][0] + c.gs[i][1] * c.gs[i][1] + c.gs[i][2] * c.gs[i][2]);
        if (!(std::fabs(nrm - 1.0) < 1e-9)) return fail(BSK_EINVAL, "wheel spin axis is not a unit vector");
        if (!(c.js[i] > 0.0)) return fail(BSK_EINVAL, "wheel inertia js must be positive");
        for (int a = 0; a < 3; ++a) {
            p.gs[i][a] = c.gs[i][a];
            for (int b = 0; b < 3; ++b) D[3 * a + b] -= c.js[i] * c.gs[i][a] * c.gs[i][b];
        }
        p.js[i] = c.js[i];
    }
    if (!inv3(D, p.dinv)) return fail(BSK_EINVAL, "hub inertia minus wheel inertia is singular");
    for (int i = 0; i < 9; ++i) { p.dmat[i] = D[i]; p.wmat[i] = c.inertia[i] - D[i]; }
    // Diagonal fast path: only when every off-diagonal of I_sc and of (I_sc - sum Js g g^T) is
    // EXACTLY zero (true for the reference's cuboid hub with the triad or the symmetric pyramid).
    diag = true;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            if (a != b && (c.inertia[3 * a + b] != 0.0 || D[3 * a + b] != 0.0)) diag = false;
    if (c.n_rw > 0) {
        // rwMotorTorque: map = CGs^T (CGs CGs^T)^-1 C,  CGs = C Gs
        double cgs[3][BSK_MAX_RW], M[9] = {0}, Mi[9];
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < c.n_rw; ++i)
                cgs[a][i] = c.ctrl_axes[3 * a] * c.gs[i][0] + c.ctrl_axes[3 * a + 1] * c.gs[i][1] +
                            c.ctrl_axes[3 * a + 2] * c.gs[i][2];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                for (int i = 0; i < c.n_rw; ++i) M[3 * a + b] += cgs[a][i] * cgs[b][i];
        if (!inv3(M, Mi)) return fail(BSK_EINVAL, "wheel set does not span the control axes");
        for (int i = 0; i < c.n_rw; ++i) {
            double t[3];
            for (int a = 0; a < 3; ++a) t[a] = cgs[0][i] * Mi[a] + cgs[1][i] * Mi[3 + a] + cgs[2][i] * Mi[6 + a];
            for (int b = 0; b < 3; ++b)
                k.map[i][b] = t[0] * c.ctrl_axes[b] + t[1] * c.ctrl_axes[3 + b] + t[2] * c.ctrl_axes[6 + b];
        }
    }
    p.f_coulomb = c.f_coulomb;
    p.fsw_every = c.fsw_every;
    p.fsw_lag = c.fsw_lag;
    p.nav_lag = c.nav_lag;
    p.req = c.req;
    p.planet_rate = c.planet_rate;
    p.sh_tab = nullptr;
    p.sh_degree = 0;
    p.sh_split = 2;
    p.sh_bodies = p.sh_bodies0 = p.sh_bodies1 = p.sh_chunk1 = 0;
    p.sh_form = 4;
    const bool full = (c.flags & (BSK_FLAG_SUN_THIRD_BODY | BSK_FLAG_DRAG | BSK_FLAG_DESAT)) != 0;
    p.ex.desat = (c.flags & BSK_FLAG_DESAT) ? 1 : 0;
    p.ex.pad_ = 0;
    k.n_thr = c.n_thr;
    k.hs_min = c.hs_min;
    k.inv_max_thrust = c.thr_max_thrust > 0.0 ? 1.0 / c.thr_max_thrust : 0.0;
    k.thr_min_fire_time = c.thr_min_fire_time;
    k.thr_min_on_time = c.thr_min_on_time;
    k.thr_max_counter = c.thr_max_counter;
    k.fsw_lag = c.fsw_lag;
    k.nav_lag = c.nav_lag;
    for (int i = 0; i < c.n_rw; ++i) { k.js[i] = c.js[i]; for (int j = 0; j < 3; ++j) k.gs[i][j] = c.gs[i][j]; }
    if (c.flags & BSK_FLAG_DESAT) {
        double dd[9] = {0}, ddi[9], Dm[BSK_MAX_THR][3];
        for (int i = 0; i < c.n_thr; ++i) {
            const double* r = c.thr_pos[i];
            const double* g = c.thr_dir[i];
            Dm[i][0] = r[1] * g[2] - r[2] * g[1]; Dm[i][1] = r[2] * g[0] - r[0] * g[2]; Dm[i][2] = r[0] * g[1] - r[1] * g[0];
            for (int j = 0; j < 3; ++j) { k.thr_f[i][j] = c.thr_max_thrust * g[j]; k.thr_l[i][j] = c.thr_max_thrust * Dm[i][j]; }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) dd[3 * a + b] += Dm[i][a] * Dm[i][b];
        }
        if (!inv3(dd, ddi)) return fail(BSK_EINVAL, "thruster set does not span the three torque axes");
        for (int i = 0; i < c.n_thr; ++i)
            for (int a = 0; a < 3; ++a) k.thr_map[i][a] = ddi[3 * a] * Dm[i][0] + ddi[3 * a + 1] * Dm[i][1] + ddi[3 * a + 2] * Dm[i][2];
    }
    p.feat = full ? bsk::FEAT_FULL : ((c.flags & BSK_FLAG_POWER) ? bsk::FEAT_POWER : bsk::FEAT_BARE);   // FEAT_FULLG: below
    if (c.flags & BSK_FLAG_LDS_SCRATCH) p.feat = bsk::FEAT_LDSS;
    p.ex.mu_sun = (c.flags & BSK_FLAG_SUN_THIRD_BODY) ? c.mu_sun : 0.0;
    p.ex.base_density = (c.flags & BSK_FLAG_DRAG) ? c.base_density : 0.0;
    p.ex.inv_scale_height = c.scale_height > 0.0 ? 1.0 / c.scale_height : 0.0;
    p.ex.inv_mass = c.mass > 0.0 ? 1.0 / c.mass : 0.0;
    p.ex.rho_skip = 1e-25;
    k.n_facets = c.n_facets;
    k.facet_axis = 1;
    for (int i = 0; i < c.n_facets && i < 8; ++i) {
        // axis-aligned normal: exactly one component is +-1, the others exactly 0
        int axis = -1, nz = 0;
        for (int j = 0; j < 3; ++j)
            if (c.facet_normal[i][j] != 0.0) { ++nz; axis = j; }
        if (nz != 1 || std::fabs(c.facet_normal[i][axis]) != 1.0) { k.facet_axis = 0; break; }
        const int sgn = c.facet_normal[i][axis] > 0.0 ? 0 : 1;
        const double acd = c.facet_area[i] * c.facet_cd[i];
        k.fa_c[sgn][axis] += acd;
        for (int j = 0; j < 3; ++j) k.fa_r[sgn][axis][j] += acd * c.facet_pos[i][j];
    }
    // half sums / half differences of the +e_k and -e_k tables (bsk_device.hpp: facet_drag)
    for (int axis = 0; axis < 3; ++axis) {
        const double cp = k.fa_c[0][axis], cm = k.fa_c[1][axis];
        k.fa_c[0][axis] = 0.5 * (cp + cm);
        k.fa_c[1][axis] = 0.5 * (cp - cm);
        for (int j = 0; j < 3; ++j) {
            const double rp = k.fa_r[0][axis][j], rm = k.fa_r[1][axis][j];
            k.fa_r[0][axis][j] = 0.5 * (rp + rm);
            k.fa_r[1][axis][j] = 0.5 * (rp - rm);
        }
    }
    if (k.facet_axis) {
        bool diagonal = true;
        for (int sgn = 0; sgn < 2; ++sgn)
            for (int axis = 0; axis < 3; ++axis)
                for (int j = 0; j < 3; ++j)
                    if (j != axis && k.fa_r[sgn][axis][j] != 0.0) diagonal = false;
        if (diagonal) k.facet_axis = 2;   // facet centres on their own normal axes: 12 table values suffice
    }
    // any other facet set with live drag runs the generic-geometry variant of the full-scenario kernel
    if (full && (c.flags & BSK_FLAG_DRAG) && c.base_density != 0.0 && k.facet_axis != 2) p.feat = bsk::FEAT_FULLG;
    for (int i = 0; i < 8; ++i) {
        k.facet_acd[i] = c.facet_area[i] * c.facet_cd[i];
        for (int j = 0; j < 3; ++j) { k.facet_n[i][j] = c.facet_normal[i][j]; k.facet_r[i][j] = c.facet_pos[i][j]; }
    }
    {
        const double AU = 149597870700.0, RSUN = 695000.0e3;
        for (int i = 0; i < 3; ++i) { p.pc.nB[i] = c.panel_normal[i]; p.pc.sun_r0[i] = c.sun_r0[i]; p.pc.sun_v[i] = c.sun_v[i]; }
        p.pc.kflux = c.panel_area * c.panel_efficiency * c.solar_flux * AU * AU;
        p.pc.draw = c.power_draw;
        p.pc.cap = c.storage_capacity;
        p.pc.req = c.req;
        p.pc.rsun = RSUN;
        p.pc.rs_plus = RSUN + c.req;
        p.pc.rs_minus = RSUN - c.req;
    }
    k.u_max = c.u_max;
    k.u_min = c.u_min;
    k.K = c.K;
    k.P = c.P;
    std::memcpy(p.obs.sigma_R0N, c.sigma_R0N, sizeof p.obs.sigma_R0N);
    std::memcpy(k.sigma_R0N, c.sigma_R0N, sizeof k.sigma_R0N);
    p.obs.inv_wheel_limit = 1.0 / c.wheel_limit;
    p.obs.charge_scale = 1.0 / 3600.0 / c.power_max;
    p.obs.reward_mult = c.reward_mult;
    p.obs.failure_penalty = c.failure_penalty;
    p.obs.r_min2 = c.r_min * c.r_min;
    p.obs.max_length = c.max_length;
    p.obs.pad_ = 0;
    // broadcast table of the full-scenario kernels (bsk_device.hpp: KTab, KA_* / KB_* / KC_*)
    for (int i = 0; i < c.n_rw; ++i) {
        for (int j = 0; j < 3; ++j) k.kt[bsk::KA_G + 3 * i + j] = c.gs[i][j];
        k.kt[bsk::KA_JS + i] = c.js[i];
        k.kt[16 + bsk::KB_IJS + i] = 1.0 / c.js[i];
        for (int j = 0; j < 3; ++j) k.kt[48 + bsk::KD_JG + 3 * i + j] = c.js[i] * c.gs[i][j];
        k.kt[48 + bsk::KD_HIJS + i] = c.dt / c.js[i];
    }
    for (int sgn = 0; sgn < 2; ++sgn)
        for (int axis = 0; axis < 3; ++axis) {
            k.kt[16 + bsk::KB_FAC + 3 * sgn + axis] = k.fa_c[sgn][axis] * p.ex.inv_mass;   // area table carries 1/m
            k.kt[16 + bsk::KB_FAD + 3 * sgn + axis] = k.fa_r[sgn][axis][axis];
        }
    k.kt[32 + bsk::KC_IMASS] = p.ex.inv_mass;
    for (int j = 0; j < 3; ++j) k.kt[32 + bsk::KC_NB + j] = c.panel_normal[j];
    k.kt[32 + bsk::KC_KFLUX] = p.pc.kflux;
    k.kt[32 + bsk::KC_RHO0] = p.ex.base_density;
    k.kt[32 + bsk::KC_NIH] = -p.ex.inv_scale_height;
    k.kt[32 + bsk::KC_REQIH] = c.req * p.ex.inv_scale_height;
    k.kt[32 + bsk::KC_RSKIP] = p.ex.rho_skip;
    k.kt[32 + bsk::KC_LOG2E] = 1.4426950408889634074;
    k.kt[32 + bsk::KC_I6] = 1.0 / 6.0; k.kt[32 + bsk::KC_I24] = 1.0 / 24.0; k.kt[32 + bsk::KC_I120] = 1.0 / 120.0;   // Atmo::advance
    k.kt[32 + bsk::KC_I720] = 1.0 / 720.0;
    {   // row E: rho0 / k!, k = 0..13 (bsk_device.hpp: atmosphere_density), -ln2 split in two parts
        long double f = 1.0L;
        for (int i = 0; i < 14; ++i) {
            if (i > 1) f *= (long double)i;
            k.kt[64 + bsk::KE_POLY + i] = (double)((long double)p.ex.base_density / f);
        }
        k.kt[64 + bsk::KE_NLN2HI] = -6.93147180369123816490e-01;
        k.kt[64 + bsk::KE_NLN2LO] = -1.90821492927058770002e-10;
    }
    // thruster subset table: row m = sums over the set bits of m, ascending thruster index
    for (int m = 0; m < (1 << BSK_MAX_THR); ++m) {
        double f[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < c.n_thr && i < BSK_MAX_THR; ++i)
            if (m & (1 << i))
                for (int j = 0; j < 3; ++j) { f[j] += k.thr_f[i][j]; f[3 + j] += k.thr_l[i][j]; }
        for (int j = 0; j < 6; ++j) k.thr_tab[m][j] = f[j];
    }
    return BSK_OK;
}

}  // namespace

struct bsk_handle {
    bsk_config cfg;
    bsk::StepParams sp;
    bsk::ColdCfg cold;
    bool diag = false;
    bsk::ColdCfg* d_cold = nullptr;
    int n = 0, nf = 0, device = 0, block = 64;
    int64_t stride = 0;      // of the state slab's field rows (padded: bsk_create)
    int64_t ostride = 0;     // of the observation / terminal-observation rows and the size of every per-env array
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double* d_state = nullptr;
    int2* d_cnt = nullptr;
    int* d_act = nullptr;
    double* d_obs = nullptr;
    double* d_reward = nullptr;
    unsigned long long* d_done_mask = nullptr;
    unsigned char* d_reason = nullptr;
    double* d_stat_sum = nullptr;
    long long* d_stat_done = nullptr;
    double* d_wave_sum = nullptr;              // stats_kernel scratch: one reward sum per 64 envs
    unsigned* d_done_part = nullptr;           // stats_kernel scratch: finished envs per first-level workgroup
    // masked-reset staging
    double* d_ic_stage = nullptr;
    int* d_idx_stage = nullptr;
    unsigned char* d_mask_stage = nullptr;
    size_t stage_cap = 0;
    double* d_sh_tab = nullptr;    // scalar-load stream (form 1)
    double* d_sh_tab4 = nullptr;   // DPP-broadcast stream (forms 4 and 5, default)
    double* d_pool = nullptr;
    double* d_term_obs = nullptr;
    int* d_episodes = nullptr;
    int n_pool = 0, pool_cap = 0;
    // profiling
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    int ev_stride = 1, ev_seq = 0;
    hipEvent_t ev_warm[2] = {nullptr, nullptr};
    bool prof = false;
    double sim_time = 0.0;
    unsigned env_base = 0;   // global index of env 0 (bsk_set_env_base)
    // device-resident surface (BSK_FLAG_EPISODE_STATS / BSK_FLAG_OBS_ROWMAJOR)
    double* d_ep_return = nullptr;
    double* d_term_return = nullptr;
    int* d_term_len = nullptr;
    unsigned char* d_done = nullptr;
    double* d_obs_rm = nullptr;
    unsigned long long* d_dbg = nullptr;   // one word per wave for probe builds (bsk_probes.hpp)
    unsigned long long* d_seal = nullptr;   // [3] what env 0's counters were behind the last reset entry point (bsk_aux.hip: stats_sealed)
    double* d_stats2 = nullptr;   // {sum of rewards, number of done envs} of the last step, as two doubles (all-reduce operand)
    bool stats_fresh = false;     // d_stat_sum / d_stat_done / d_stats2 hold the LAST STEP's batch scalars (snapshot_stats)
    bool step_stats = false;      // bsk_set_step_stats: step launches write d_wave_sum themselves (a request = the join kernel alone)
    bool wave_sums_fresh = false; // ... and the last launch that wrote rewards did so
    bool stepped = false;         // some step has run since the handle was created
    // A launch of this handle has been recorded into a HIP graph (note_capture): replays advance the device without this
    // host-side state, so from then on nothing evaluated at enqueue time is trusted - the batch scalars are formed again
    // whenever asked for (stats_fresh ignored) and the bare levels read the battery charge again (static_charge off).
    bool replayable = false;
    // error word the kernels can raise (page-locked host memory, device-visible): checked by every synchronising entry point
    int* h_err = nullptr;
    // bare levels: no spacecraft of the batch / of the reset pool started its episode with an empty battery (bsk_launch.hpp:
    // StepArgs::static_charge).  Known after a reset of the whole batch; withdrawn by bsk_set_state until the next one.
    bool charge_pos = false, pool_charge_pos = false;
    // pair form of the step kernel (bsk_device.hpp: PairLds): used for launches of >= pair_min_substeps sub-steps of batches
    // of <= pair_max_envs spacecraft where it is built (power / full-scenario levels, point mass or J2, diagonal hub).  Measured
    // (profiles/r03/pair_form.txt): -13 % per env step up to one pair per CU (16 384 spacecraft), level with the single-wave
    // form up to three pairs per CU, 7 % slower at four (65 536).  BSKGPU_PAIR=0 / 1 forces it off / on for every launch.
    bool pair_ok = false, last_pair = false;
    int pair_min_substeps = 16, pair_max_envs = 16384;
    // three-wave form (bsk_device.hpp: TriX): the pair form with the dynamics wave cut into a translational and a rotational
    // wave; full-scenario level only, preferred over the pair form where both apply (profiles/r03/tri_form.txt: -16 % against
    // the pair form up to one workgroup per CU, twice the time above).  BSKGPU_TRI=0 / 1 forces it off / on.
    bool tri_ok = false, last_tri = false;
    bool last_rollout = false, last_rollout_act = false;    // the last launch was bsk_step_n's rollout kernel (with per-step actions); bsk_kernel_info
    int tri_min_substeps = 16, tri_max_envs = 16384;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int validate(const bsk_config& c) {
    if (c.abi_version != BSK_ABI_VERSION || c.struct_size != sizeof(bsk_config))
        return fail(BSK_EABI, "bsk_config abi_version/struct_size mismatch (header " + std::to_string(BSK_ABI_VERSION) +
                                  "/" + std::to_string(sizeof(bsk_config)) + ")");
    if (!(c.dt > 0.0)) return fail(BSK_EINVAL, "dt must be positive");
    if (c.fsw_every < 1 || c.fsw_every > 2047) return fail(BSK_EINVAL, "fsw_every must be in 1..2047");
    if (c.max_length < 0 || c.max_length > 1000000) return fail(BSK_EINVAL, "max_length must be in 0..1000000");
    if (c.fsw_lag != 0 && c.fsw_lag != 1) return fail(BSK_EINVAL, "fsw_lag must be 0 or 1");
    if (c.nav_lag != 0 && c.nav_lag != 1) return fail(BSK_EINVAL, "nav_lag must be 0 or 1");
    if (c.n_rw != 0 && c.n_rw != 3 && c.n_rw != 4) return fail(BSK_EINVAL, "n_rw must be 0, 3 or 4");
    if (c.gravity_model != BSK_GRAV_PM && c.gravity_model != BSK_GRAV_PM_J2 && c.gravity_model != BSK_GRAV_SH)
        return fail(BSK_EINVAL, "unknown gravity_model");
    if (c.gravity_model == BSK_GRAV_SH && (c.sh_degree < 2 || c.sh_degree > BSK_MAX_SH_DEGREE))
        return fail(BSK_EINVAL, "sh_degree must be in 2..70 for BSK_GRAV_SH");
    if ((c.flags & (BSK_FLAG_SUN_THIRD_BODY | BSK_FLAG_DRAG | BSK_FLAG_DESAT)) && !(c.flags & BSK_FLAG_POWER))
        return fail(BSK_EINVAL, "BSK_FLAG_SUN_THIRD_BODY / BSK_FLAG_DRAG / BSK_FLAG_DESAT are built in the full-scenario kernel: set BSK_FLAG_POWER too");
    if ((c.flags & BSK_FLAG_DESAT) && (c.n_thr < 3 || c.n_thr > BSK_MAX_THR || c.n_rw == 0 || !(c.thr_max_thrust > 0.0) || !(c.mass > 0.0)))
        return fail(BSK_EINVAL, "BSK_FLAG_DESAT needs 3..8 thrusters, wheels, thr_max_thrust > 0 and mass > 0");
    if ((c.flags & BSK_FLAG_DRAG) && (c.n_facets < 0 || c.n_facets > 8 || !(c.scale_height > 0.0) || !(c.mass > 0.0)))
        return fail(BSK_EINVAL, "BSK_FLAG_DRAG needs 0..8 facets, scale_height > 0 and mass > 0");
    if ((c.flags & BSK_FLAG_LDS_SCRATCH) && ((c.flags & BSK_FLAG_POWER) || c.gravity_model == BSK_GRAV_SH))
        return fail(BSK_EINVAL, "BSK_FLAG_LDS_SCRATCH is built for the bare propagator (point mass / J2, no power system) only");
    if (!(c.mu > 0.0) || !(c.req > 0.0)) return fail(BSK_EINVAL, "mu and req must be positive");
    if (!(c.wheel_limit > 0.0) || !(c.power_max > 0.0)) return fail(BSK_EINVAL, "wheel_limit and power_max must be positive");
    if ((c.flags & BSK_FLAG_POWER) && !(c.storage_capacity > 0.0 && c.sun_r0[0] * c.sun_r0[0] + c.sun_r0[1] * c.sun_r0[1] + c.sun_r0[2] * c.sun_r0[2] > 0.0))
        return fail(BSK_EINVAL, "BSK_FLAG_POWER needs storage_capacity > 0 and a Sun position");
    return BSK_OK;
}

// Extra elements per field row of the state slab.  An EMPIRICAL constant, not a derived one: with the slab's rows an odd multiple of
// 256 B apart the K = 1 launch of 65 536 spacecraft is 2 - 3 % shorter than with rows at a power-of-two distance (6.20 against 6.36 us
// wall per launch; profiles/r05/stride_pad.txt, ten alternations in stride_pad_ab.txt), a micro-benchmark of the bare access pattern
// does not reproduce it (tools/micro/row_channels.hip) and the mechanism is not established.  Applied only over the range of batch
// sizes it was measured to help at (profiles/r06/stride_pad.txt, one box, alternating: 65 536 -2.7 %, 98 304 -0.5 %; 32 768 and
// 131 072 nothing either way, 4 Mi slightly slower); everywhere else the rows are N rounded up to 256 like every other per-env
// array of the handle.
#ifndef BSK_TUNABLES
#define BSK_TUNABLES 0
#endif
constexpr int SLAB_PAD_ELEMS = 32, SLAB_PAD_MIN_ENVS = 65536, SLAB_PAD_MAX_ENVS = 98304;
int slab_row_pad(int n_envs) { return (n_envs >= SLAB_PAD_MIN_ENVS && n_envs <= SLAB_PAD_MAX_ENVS) ? SLAB_PAD_ELEMS : 0; }

int ensure_stage(bsk_handle* h, size_t m) {
    if (m <= h->stage_cap) return BSK_OK;
    if (h->d_ic_stage) (void)hipFree(h->d_ic_stage);
    if (h->d_idx_stage) (void)hipFree(h->d_idx_stage);
    h->d_ic_stage = nullptr;
    h->d_idx_stage = nullptr;
    h->stage_cap = 0;
    HIP_TRY(hipMalloc(&h->d_ic_stage, m * h->nf * sizeof(double)));
    HIP_TRY(hipMalloc(&h->d_idx_stage, m * sizeof(int)));
    h->stage_cap = m;
    return BSK_OK;
}

bsk::ResetOut reset_out(const bsk_handle* h) {
    bsk::ResetOut ro;
    ro.obs = h->d_obs; ro.ostride = h->ostride; ro.obs_rm = h->d_obs_rm; ro.reward = h->d_reward; ro.reason = h->d_reason; ro.done = h->d_done;
    ro.ep_return = h->d_ep_return;
    ro.inv_wheel_limit = h->sp.obs.inv_wheel_limit; ro.charge_scale = h->sp.obs.charge_scale;
    ro.n_rw = h->cfg.n_rw;
    return ro;
}

// The batch scalars of the last step (sum of rewards, number of done envs) are formed from the reward buffer and the done
// ballots by a kernel of their own, once, when somebody asks - or just before a reset entry point overwrites the restarted
// envs' rewards with zeros (the vec env auto-resets BEFORE a training loop reads the step's statistics).
static bool note_capture(bsk_handle* h) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!h->replayable && hipStreamIsCapturing(h->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) h->replayable = true;
    return h->replayable;
}

static bsk::StatsSeal stats_seal(const bsk_handle* h) {
    return bsk::StatsSeal{(const unsigned long long*)h->d_cnt, h->d_episodes, h->d_seal};
}

static int snapshot_stats(bsk_handle* h) {
    const bool replayable = note_capture(h);
    if (h->stats_fresh && !replayable) return BSK_OK;
    // (a request that is itself being captured must not freeze "the last launch wrote the wave sums" into the graph: a replay may
    // follow steps that did not - the two-level form whenever the handle is replayable)
    HIP_TRY(bsk::launch_stats(h->d_reward, h->n, h->d_done_mask, (h->n + 63) / 64, h->d_wave_sum, h->d_done_part,
                              h->d_stat_sum, h->d_stat_done, h->d_stats2, h->wave_sums_fresh && !replayable, stats_seal(h), h->stream));
    h->stats_fresh = true;
    return BSK_OK;
}

// A reset entry point has taken the snapshot and enqueued the kernels that zero the restarted envs' rewards: from here until the
// next step launch the snapshot is what bsk_get_batch_stats* report, also on a handle whose launches replay from a HIP graph
// (bsk_aux.hip: stats_sealed).
static int seal_stats(bsk_handle* h) {
    if (!h->stepped) return BSK_OK;
    HIP_TRY(bsk::launch_seal(stats_seal(h), h->stream));
    return BSK_OK;
}

// After a stream synchronisation: has a kernel raised the handle's error word?  (bsk_device.hpp: BSK_DEVERR_*)
int check_device_error(bsk_handle* h) {
    if (!h->h_err) return BSK_OK;
    const int e = *(volatile int*)h->h_err;
    if (e == 0) return BSK_OK;
    *(volatile int*)h->h_err = 0;
    if (e == bsk::BSK_DEVERR_TRI_EXCHANGE)
        return fail(BSK_EHIP, "step kernel (three-wave form): a wave waited 2^20 polls for its partner's stage value and gave up; "
                              "the results of that launch are invalid (observations were set to NaN)");
    return fail(BSK_EHIP, "step kernel raised device error " + std::to_string(e));
}
#define SYNC_CHECKED(h) do { HIP_SYNC(hipStreamSynchronize((h)->stream)); int rc_ = check_device_error(h); if (rc_) return rc_; } while (0)

void fill_buffers(bsk_handle* h, bsk::StepBuffers& b, const void* d_actions, int substeps, int act_shift, bool static_charge) {
    b.cold = h->d_cold;
    b.st = h->d_state;
    b.cnt = h->d_cnt;
    b.act = (const int*)d_actions;
    b.act_shift = act_shift;
    b.static_charge = static_charge ? 1 : 0;
    b.ep_return = h->d_ep_return; b.term_return = h->d_term_return; b.term_len = h->d_term_len; b.done = h->d_done;
    b.obs_rm = h->d_obs_rm; b.err = h->h_err; b.dbg = h->d_dbg;
    // (above 2 Mi spacecraft one workgroup joining 32 768+ wave sums AND as many done ballots is no faster than the two-level form; a
    // handle whose launches have been captured keeps the two-level form too: "the last launch wrote the wave sums" is host-side
    // knowledge, and a replayed graph steps without telling the host)
    b.wave_sum = (h->step_stats && !h->replayable && h->n <= (1 << 21)) ? h->d_wave_sum : nullptr;
    b.obs = h->d_obs;
    b.reward = h->d_reward;
    b.done_mask = h->d_done_mask;
    b.reason = h->d_reason;
    b.stride = h->stride;
    b.ostride = h->ostride;
    b.n = h->n;
    b.substeps = substeps;
    b.pool = h->d_pool;
    b.term_obs = h->d_term_obs;
    b.episodes = h->d_episodes;
    b.n_pool = h->n_pool;
    b.n_fields = h->nf;
    b.env_base = h->env_base;
}

// Dispatch-timestamp sampling.  stride == 1: every launch is stamped.  stride > 1: launches
// seq % stride == 0 and 1 are stamped as a pair and only the second is counted — the first one
// absorbs the transition from un-stamped back-to-back launches (a lone stamped launch reads ~25 %
// long), the second runs under the same conditions as in an every-launch-stamped run.
int stamp_events(bsk_handle* h, hipEvent_t& e0, hipEvent_t& e1) {
    e0 = e1 = nullptr;
    if (!h->prof) return BSK_OK;
    const int ph = h->ev_seq++ % h->ev_stride;
    if (h->ev_stride == 1 || ph == 1) {
        if (h->ev_used + 2 <= (int)h->ev.size()) {
            e0 = h->ev[h->ev_used];
            e1 = h->ev[h->ev_used + 1];
            h->ev_used += 2;
        }
    } else if (ph == 0) {
        if (!h->ev_warm[0]) {
            HIP_TRY(hipEventCreate(&h->ev_warm[0]));
            HIP_TRY(hipEventCreate(&h->ev_warm[1]));
        }
        e0 = h->ev_warm[0];
        e1 = h->ev_warm[1];
    }
    return BSK_OK;
}

int do_step(bsk_handle* h, const void* d_actions, int substeps, int act_shift) {
    if (h->cfg.gravity_model == BSK_GRAV_SH && !h->sp.sh_tab)
        return fail(BSK_EINVAL, "BSK_GRAV_SH: call bsk_set_gravity_sh before stepping");
    if ((h->cfg.flags & BSK_FLAG_AUTO_RESET) && h->n_pool == 0)
        return fail(BSK_EINVAL, "BSK_FLAG_AUTO_RESET: call bsk_set_ic_pool before stepping");
    bsk::StepBuffers b;
    const bool replayable = note_capture(h);      // (a captured launch must not freeze a host-side decision into the graph)
    fill_buffers(h, b, d_actions, substeps, act_shift,
                 !replayable && (h->sp.feat == bsk::FEAT_BARE || h->sp.feat == bsk::FEAT_LDSS) && h->charge_pos && (h->n_pool == 0 || h->pool_charge_pos));
    hipEvent_t e0, e1;
    { int rc = stamp_events(h, e0, e1); if (rc) return rc; }
    h->sp.tri = (h->tri_ok && substeps >= h->tri_min_substeps && h->n <= h->tri_max_envs) ? 1 : 0;
    h->sp.pair = (!h->sp.tri && h->pair_ok && substeps >= h->pair_min_substeps && h->n <= h->pair_max_envs) ? 1 : 0;
    h->last_pair = h->sp.pair != 0;
    h->last_tri = h->sp.tri != 0;
    h->last_rollout = false;
    HIP_TRY(bsk::launch_step(h->cfg.gravity_model, h->cfg.n_rw, h->diag, h->sp.feat, h->sp, b, h->block, h->stream, e0, e1));
    h->stats_fresh = false;
    h->wave_sums_fresh = b.wave_sum != nullptr;
    h->stepped = true;
    return BSK_OK;
}

}  // namespace

static int ensure_pool_buffers(bsk_handle* h, int n_pool);

extern "C" {

const char* bsk_last_error(void) { return g_err.c_str(); }
const char* bsk_version(void) { return "bskgpu 0.1 (gfx950)"; }

int bsk_default_config(bsk_config* c, int n_rw, int gravity_model) {
    if (!c) return fail(BSK_EINVAL, "cfg is NULL");
    if (n_rw != 0 && n_rw != 3 && n_rw != 4) return fail(BSK_EINVAL, "n_rw must be 0, 3 or 4");
    std::memset(c, 0, sizeof *c);
    c->abi_version = BSK_ABI_VERSION;
    c->struct_size = sizeof *c;
    c->dt = 0.1;
    c->fsw_every = 10;
    c->gravity_model = gravity_model;
    c->sh_degree = 0;
    c->n_rw = n_rw;
    c->flags = 0;
    c->max_length = 540;
    c->fsw_lag = 1;
    c->nav_lag = 1;
    c->mu = 0.3986004415e15;
    c->req = 6378136.6;
    c->j2 = std::sqrt(5.0) * 4.841693e-4;
    c->planet_rate = 7.2921159e-5;
    const double m = 330.0, w = 1.38, dpt = 1.04, ht = 1.58;
    c->mass = m;
    c->inertia[0] = 1. / 12. * m * (w * w + dpt * dpt);
    c->inertia[4] = 1. / 12. * m * (dpt * dpt + ht * ht);
    c->inertia[8] = 1. / 12. * m * (w * w + ht * ht);
    const double D2R = M_PI / 180.0;
    if (n_rw == 3) {
        for (int i = 0; i < 3; ++i) c->gs[i][i] = 1.0;
    } else if (n_rw == 4) {
        // one quadrant's components with explicit signs: exactly symmetric set (see
        // actuatorPrimatives.balancedHR16Pyramid), so sum(g g^T) is exactly diagonal
        const double el = 40.0 * D2R, az = 45.0 * D2R;
        double cx = std::cos(az) * std::cos(el), cy = std::sin(az) * std::cos(el), cz = std::sin(el);
        const double nn = std::sqrt(cx * cx + cy * cy + cz * cz);
        cx /= nn; cy /= nn; cz /= nn;
        const int sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1};
        for (int i = 0; i < 4; ++i) {
            c->gs[i][0] = sx[i] * cx;
            c->gs[i][1] = sy[i] * cy;
            c->gs[i][2] = cz;
        }
    }
    for (int i = 0; i < n_rw; ++i) c->js[i] = 50.0 / (6000.0 * M_PI * 2.0 / 60.0);
    c->u_max = 0.2;
    c->u_min = 0.00001;
    c->f_coulomb = 0.0005;
    c->K = 7.0;
    c->P = 35.0;
    c->sigma_R0N[0] = 1.0;
    c->ctrl_axes[0] = c->ctrl_axes[4] = c->ctrl_axes[8] = 1.0;
    c->wheel_limit = 3000.0 * (2.0 * M_PI / 60.0);
    c->power_max = 20.0;
    c->reward_mult = 1.0 / 540.0;
    c->failure_penalty = 1.0;
    c->r_min = 6378.1366 / 1000.0;
    c->panel_normal[1] = -1.0;
    c->panel_area = 0.2 * 0.3;
    c->panel_efficiency = 0.20;
    c->power_draw = -5.0;
    c->storage_capacity = 20.0 * 3600.0;
    c->solar_flux = 1372.5398;
    // epoch 2021 MAY 04 07:47:48.965 UTC (JD 2459338.5 + 07:47:48.965)
    const double jd0 = 2459338.5 + (7.0 * 3600.0 + 47.0 * 60.0 + 48.965) / 86400.0;
    double p0[3], p1[3];
    sun_position(jd0, p0);
    sun_position(jd0 + 1.0, p1);
    for (int k = 0; k < 3; ++k) {
        c->sun_r0[k] = p0[k];
        c->sun_v[k] = (p1[k] - p0[k]) / 86400.0;
    }
    c->mu_sun = 1.32712440018e20;
    c->hs_min = 4.0;
    c->thr_max_counter = 4;
    c->thr_min_fire_time = 0.002;
    {   // idealMonarc1Octet (actuatorPrimatives.py:66-161), MOOG Monarc-1: 0.9 N, MinOnTime 0.02 s
        const double x = 3.874945160902288e-2, y = 1.206182747348013, z = 0.85245, x2 = 3.8749451609022656e-2;
        const double loc[8][3] = {{x, -y, z}, {x, -y, -z}, {-x2, -y, z}, {-x2, -y, -z}, {-x, y, z}, {-x, y, -z}, {x2, y, z}, {x2, y, -z}};
        const double a = 0.7071067811865476, b = 0.7071067811865475;
        const double dir[8][3] = {{-a, b, 0}, {-a, b, 0}, {b, a, 0}, {b, a, 0}, {a, -b, 0}, {a, -b, 0}, {-b, -a, 0}, {-b, -a, 0}};
        c->n_thr = 8;
        for (int i = 0; i < 8; ++i)
            for (int k = 0; k < 3; ++k) { c->thr_pos[i][k] = loc[i][k]; c->thr_dir[i][k] = dir[i][k]; }
        c->thr_max_thrust = 0.9;
        c->thr_min_on_time = 0.020;
    }
    c->base_density = 1.22;
    c->scale_height = 8.0e3;
    // 6U cubesat facets + two 1x2 m panels, Cd 2.2 (leoPowerAttitudeSimulator.py:272-281)
    const double fa[8] = {0.2 * 0.3, 0.2 * 0.3, 0.1 * 0.2, 0.1 * 0.2, 0.1 * 0.3, 0.1 * 0.3, 1. * 2., 1. * 2.};
    const double fn[8][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, 1, 0}, {0, -1, 0}};
    const double fp[8][3] = {{0.05, 0, 0}, {0.05, 0, 0}, {0, 0.15, 0}, {0, -0.15, 0}, {0, 0, 0.1}, {0, 0, -0.1}, {0, 2., 0}, {0, 2., 0}};
    c->n_facets = 8;
    for (int i = 0; i < 8; ++i) {
        c->facet_area[i] = fa[i];
        c->facet_cd[i] = 2.2;
        for (int k = 0; k < 3; ++k) { c->facet_normal[i][k] = fn[i][k]; c->facet_pos[i][k] = fp[i][k]; }
    }
    return BSK_OK;
}

int bsk_create(const bsk_config* cfg, int n_envs, int device_id, void* stream, bsk_handle** out) {
    if (!cfg || !out) return fail(BSK_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    if (n_envs < 1 || n_envs > (1 << 28)) return fail(BSK_EINVAL, "n_envs must be in 1..2^28");
    int rc = validate(*cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(BSK_ENODEV, "no HIP device visible: libbskgpu has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(BSK_ENODEV, "device_id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BSK_ENODEV, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    DeviceGuard guard(device_id);

    bsk_handle* h = new bsk_handle();
    h->cfg = *cfg;
    rc = build_params(*cfg, h->sp, h->cold, h->diag);
    if (rc) { delete h; return rc; }
    h->n = n_envs;
    h->nf = BSK_NF_BASE + cfg->n_rw + BSK_NF_TAIL;
    h->device = device_id;
    // Rows of N rounded up to 256 elements for everything a consumer sees (observation / reward rows, per-env arrays: a shard that
    // fills its rows is one contiguous block for the exchange step); the state slab's field rows carry slab_row_pad() more.
    h->ostride = ((int64_t)n_envs + 255) / 256 * 256;
    h->stride = h->ostride + slab_row_pad(n_envs);
#if BSK_TUNABLES
    // measurement overrides, `make tunables` builds only (variants/tunables.so; the product library never reads them):
    // extra elements per observation / reward row too; the slab's extra elements per row (multiples of 16)
    if (const char* sp = std::getenv("BSKGPU_OSTRIDE_PAD")) {
        const int v = std::atoi(sp);
        if (v > 0 && v % 16 == 0) h->ostride += v;
    }
    if (const char* sp = std::getenv("BSKGPU_STRIDE_PAD")) {
        const int v = std::atoi(sp);
        if (v >= 0 && v % 16 == 0) h->stride = ((int64_t)n_envs + 255) / 256 * 256 + v;
    }
#endif
    // 64-lane workgroups spread a small batch over more CUs (65 536 envs = 1 024 waves = 4 per CU);
    // large batches use 256 so the dispatcher has fewer workgroups to place.
    h->block = n_envs >= (1 << 20) ? 256 : 64;
#if BSK_TUNABLES
    if (const char* b = std::getenv("BSKGPU_BLOCK")) {   // measurement override: 64, 128 or 256
        const int v = std::atoi(b);
        if (v == 64 || v == 128 || v == 256) h->block = v;
    }
#endif
    // the power-system kernels carry 37.6 KB of LDS per wave: one wave per workgroup at every batch size
    if (cfg->flags & BSK_FLAG_POWER) h->block = 64;
    // both wave-split forms pay while every workgroup has a CU (and its LDS) to itself: 64 spacecraft per CU of THIS device
    // (256 CUs on a whole MI355X; fewer in a partitioned mode)
    if (prop.multiProcessorCount > 0) h->pair_max_envs = h->tri_max_envs = 64 * prop.multiProcessorCount;
    h->pair_ok = bsk::pair_available(cfg->gravity_model, h->diag, h->sp.feat);
    h->sp.pair_shift = 31;     // no swap: the hardware already places one wave 0 and one wave 1 of different workgroups on a SIMD (tools/micro/placement.hip)
#if BSK_TUNABLES
    if (const char* ps = std::getenv("BSKGPU_PAIR_SHIFT")) h->sp.pair_shift = std::max(0, std::min(31, std::atoi(ps)));
#endif
    if (const char* pv = std::getenv("BSKGPU_PAIR")) {
        const int v = std::atoi(pv);
        if (v == 0) h->pair_ok = false;
        else { h->pair_min_substeps = 1; h->pair_max_envs = 1 << 28; }   // every launch (measurement / tests)
    }
    h->tri_ok = bsk::tri_available(cfg->gravity_model, h->diag, h->sp.feat);
    if (const char* tv = std::getenv("BSKGPU_TRI")) {
        const int v = std::atoi(tv);
        if (v == 0) h->tri_ok = false;
        else { h->tri_min_substeps = 1; h->tri_max_envs = 1 << 28; }     // every launch (measurement / tests)
    }
    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(BSK_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
        h->own_stream = true;
    }
    const int64_t S = h->ostride;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, h->stream);
        return e;
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = alloc((void**)&h->d_state, (size_t)h->nf * h->stride * sizeof(double));
    if (e == hipSuccess) e = alloc((void**)&h->d_cnt, (size_t)S * sizeof(int2));
    if (e == hipSuccess) e = alloc((void**)&h->d_act, (size_t)S * sizeof(int));
    // observation rows and the reward row in ONE allocation, f64[6][stride]: a shard whose size equals its stride hands the
    // exchange step one contiguous block per rank (rccl.py: rank-major gather)
    if (e == hipSuccess) e = alloc((void**)&h->d_obs, (size_t)6 * S * sizeof(double));
    if (e == hipSuccess) h->d_reward = h->d_obs + (size_t)5 * S;
    if (e == hipSuccess) e = alloc((void**)&h->d_done_mask, (size_t)(S / 64) * sizeof(unsigned long long));
    if (e == hipSuccess) e = alloc((void**)&h->d_reason, (size_t)S);
    if (e == hipSuccess) e = alloc((void**)&h->d_stat_sum, sizeof(double));
    if (e == hipSuccess) e = alloc((void**)&h->d_stat_done, sizeof(long long));
    if (e == hipSuccess) e = alloc((void**)&h->d_stats2, 2 * sizeof(double));
    if (e == hipSuccess) e = alloc((void**)&h->d_seal, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = alloc((void**)&h->d_wave_sum, (size_t)(S / 64) * sizeof(double));
    if (e == hipSuccess) e = alloc((void**)&h->d_done_part, (size_t)bsk::stats_done_parts() * sizeof(unsigned));
    if (e == hipSuccess) e = alloc((void**)&h->d_dbg, (size_t)(S / 64) * sizeof(unsigned long long));
    if (e == hipSuccess && (cfg->flags & BSK_FLAG_EPISODE_STATS)) {
        e = alloc((void**)&h->d_ep_return, (size_t)S * sizeof(double));
        if (e == hipSuccess) e = alloc((void**)&h->d_term_return, (size_t)S * sizeof(double));
        if (e == hipSuccess) e = alloc((void**)&h->d_term_len, (size_t)S * sizeof(int));
        if (e == hipSuccess) e = alloc((void**)&h->d_done, (size_t)S);
    }
    if (e == hipSuccess && (cfg->flags & BSK_FLAG_OBS_ROWMAJOR)) e = alloc((void**)&h->d_obs_rm, (size_t)5 * S * sizeof(double));
    if (e == hipSuccess) {
        e = hipHostMalloc((void**)&h->h_err, sizeof(int), hipHostMallocDefault);
        if (e == hipSuccess) *h->h_err = 0;
    }
    if (e == hipSuccess) e = alloc((void**)&h->d_cold, sizeof(bsk::ColdCfg));
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_cold, &h->cold, sizeof(bsk::ColdCfg), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        int code = fail(e == hipErrorOutOfMemory ? BSK_ENOMEM : BSK_EHIP, std::string("device allocation: ") + hipGetErrorString(e));
        bsk_destroy(h);
        return code;
    }
    *out = h;
    return BSK_OK;
}

void bsk_destroy(bsk_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t ev : h->ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->ev_warm)
        if (ev) (void)hipEventDestroy(ev);
    void* bufs[] = {h->d_state, h->d_cnt, h->d_act, h->d_obs /* + d_reward: one block */, h->d_done_mask, h->d_reason,
                    h->d_stat_sum, h->d_stat_done, h->d_ic_stage, h->d_idx_stage, h->d_mask_stage, h->d_cold, h->d_sh_tab, h->d_sh_tab4, h->d_pool, h->d_term_obs, h->d_episodes,
                    h->d_ep_return, h->d_term_return, h->d_term_len, h->d_done, h->d_obs_rm, h->d_stats2, h->d_dbg, h->d_wave_sum, h->d_done_part, h->d_seal};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (h->h_err) (void)hipHostFree(h->h_err);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int bsk_set_gravity_sh(bsk_handle* h, int degree, const double* cbar, const double* sbar) {
    if (!h || !cbar || !sbar) return fail(BSK_EINVAL, "handle/cbar/sbar is NULL");
    if (h->cfg.gravity_model != BSK_GRAV_SH) return fail(BSK_EINVAL, "handle was not created with BSK_GRAV_SH");
    if (degree != h->cfg.sh_degree) return fail(BSK_EINVAL, "degree differs from bsk_config.sh_degree");
    if (cbar[0] != 1.0) return fail(BSK_EINVAL, "Cbar[0][0] must be 1 (normalised coefficients)");
    DeviceGuard guard(h->device);
    std::vector<double> tab, tab4;
    build_sh_table(degree, cbar, sbar, tab);
    const ShLayout lay = build_sh_table_dpp(degree, cbar, sbar, tab4);
    HIP_SYNC(hipStreamSynchronize(h->stream));
    if (h->d_sh_tab) { (void)hipFree(h->d_sh_tab); h->d_sh_tab = nullptr; }
    if (h->d_sh_tab4) { (void)hipFree(h->d_sh_tab4); h->d_sh_tab4 = nullptr; }
    HIP_TRY(hipMalloc(&h->d_sh_tab, tab.size() * sizeof(double)));
    HIP_COPY(hipMemcpy(h->d_sh_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_sh_tab4, tab4.size() * sizeof(double)));
    HIP_COPY(hipMemcpy(h->d_sh_tab4, tab4.data(), tab4.size() * sizeof(double), hipMemcpyHostToDevice));
    h->sp.sh_degree = degree;
    h->sp.sh_split = lay.split;
    h->sp.sh_chunk1 = lay.chunk1;
    h->sp.sh_bodies = lay.bodies;
    h->sp.sh_bodies0 = lay.bodies0;
    h->sp.sh_bodies1 = lay.bodies1;
    // Form of the harmonics kernel.  Below two 64-lane waves per SIMD (1 024 SIMDs on MI355X) each
    // spacecraft's walk is split over two cooperating waves (form 5), above that one wave walks it
    // (form 4); the two give bit-identical results.  BSKGPU_SH_FORM=1|4|5 forces a form (measurement).
    int n_cu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    }
    h->sp.sh_form = (h->n < 2 * (4 * n_cu) * 64) ? 5 : 4;
    if (const char* f = std::getenv("BSKGPU_SH_FORM")) {
        const int v = std::atoi(f);
        if (v == 1 || v == 4 || v == 5) h->sp.sh_form = v;
    }
    h->sp.sh_tab = h->sp.sh_form == 1 ? h->d_sh_tab : h->d_sh_tab4;
    return BSK_OK;
}

int bsk_n_fields(const bsk_handle* h) { return h ? h->nf : BSK_EINVAL; }

int bsk_reset(bsk_handle* h, const uint8_t* mask, const double* ic) {
    if (!h || !ic) return fail(BSK_EINVAL, "handle/ic is NULL");
    DeviceGuard guard(h->device);
    if (h->stepped) { int rc = snapshot_stats(h); if (rc) return rc; }   // the last step's batch scalars, before its rewards are overwritten
    const size_t row = (size_t)h->n * sizeof(double);
    const double* ic_charge = ic + (size_t)(BSK_NF_BASE + h->cfg.n_rw + BSK_T_CHARGE) * h->n;
    if (!mask) {
        h->charge_pos = true;
        for (int i = 0; i < h->n; ++i) h->charge_pos = h->charge_pos && ic_charge[i] > 0.0;
        HIP_COPY(hipMemcpy2DAsync(h->d_state, (size_t)h->stride * sizeof(double), ic, row, row, h->nf,
                                 hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(h->d_cnt, 0, (size_t)h->ostride * sizeof(int2), h->stream));
        HIP_TRY(bsk::launch_init_outputs(h->d_state, h->stride, nullptr, h->n, reset_out(h), h->stream));
        { int rc = seal_stats(h); if (rc) return rc; }
        HIP_SYNC(hipStreamSynchronize(h->stream));
        return BSK_OK;
    }
    std::vector<int> idx;
    for (int i = 0; i < h->n; ++i)
        if (mask[i]) { idx.push_back(i); h->charge_pos = h->charge_pos && ic_charge[i] > 0.0; }
    const size_t m = idx.size();
    if (m == 0) return BSK_OK;
    std::vector<double> compact(m * h->nf);
    for (int f = 0; f < h->nf; ++f)
        for (size_t t = 0; t < m; ++t) compact[(size_t)f * m + t] = ic[(size_t)f * h->n + idx[t]];
    int rc = ensure_stage(h, m);
    if (rc) return rc;
    HIP_COPY(hipMemcpyAsync(h->d_ic_stage, compact.data(), compact.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_COPY(hipMemcpyAsync(h->d_idx_stage, idx.data(), m * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(bsk::launch_scatter_reset(h->d_state, h->stride, h->nf, h->d_ic_stage, h->d_idx_stage, (int)m, h->d_cnt, h->stream));
    HIP_TRY(bsk::launch_init_outputs(h->d_state, h->stride, h->d_idx_stage, (int)m, reset_out(h), h->stream));
    { int rc = seal_stats(h); if (rc) return rc; }
    HIP_SYNC(hipStreamSynchronize(h->stream));
    return BSK_OK;
}

int bsk_step(bsk_handle* h, const int32_t* actions, int substeps) {
    if (!h || !actions) return fail(BSK_EINVAL, "handle/actions is NULL");
    if (substeps < 1) return fail(BSK_EINVAL, "substeps must be >= 1");
    DeviceGuard guard(h->device);
    HIP_COPY(hipMemcpyAsync(h->d_act, actions, (size_t)h->n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return do_step(h, h->d_act, substeps, 1);
}

int bsk_step_device(bsk_handle* h, const int32_t* d_actions, int substeps) {
    if (!h || !d_actions) return fail(BSK_EINVAL, "handle/actions is NULL");
    if (substeps < 1) return fail(BSK_EINVAL, "substeps must be >= 1");
    DeviceGuard guard(h->device);
    return do_step(h, d_actions, substeps, 1);
}

int bsk_step_device_i64(bsk_handle* h, const int64_t* d_actions, int substeps) {
    if (!h || !d_actions) return fail(BSK_EINVAL, "handle/actions is NULL");
    if (substeps < 1) return fail(BSK_EINVAL, "substeps must be >= 1");
    DeviceGuard guard(h->device);
    return do_step(h, d_actions, substeps, 0);       // the kernel reads the low word of every little-endian int64
}

int bsk_step_n(bsk_handle* h, const int32_t* d_actions, int32_t constant_action, int substeps, int n_steps,
               double* d_obs_hist, double* d_reward_hist, uint8_t* d_reason_hist) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (substeps < 1 || n_steps < 1) return fail(BSK_EINVAL, "substeps and n_steps must be >= 1");
    if (!d_actions && (constant_action < 0 || constant_action > 2)) return fail(BSK_EINVAL, "constant_action must be 0, 1 or 2");
    if ((h->cfg.flags & BSK_FLAG_AUTO_RESET) && h->n_pool == 0)
        return fail(BSK_EINVAL, "BSK_FLAG_AUTO_RESET: call bsk_set_ic_pool before stepping");
    DeviceGuard guard(h->device);
    if (!bsk::rollout_available(h->cfg.gravity_model, h->sp.feat)) {
        // The scenario levels, the harmonics, the LDS-scratch level: the env steps stay separate launches of step_kernel (at the
        // reference's 1 800 sub-steps per env step a launch's overhead is 0.2 % of the step; the fused kernel exists where it is not),
        // each followed by a row of the history - all enqueued here, no host visit in between.  Same results by construction.
        if (!d_actions) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)h->d_act, constant_action, (size_t)h->n, h->stream));
        for (int t = 0; t < n_steps; ++t) {
            int rc = do_step(h, d_actions ? (const void*)(d_actions + (size_t)t * h->n) : (const void*)h->d_act, substeps, 1);
            if (rc) return rc;
            HIP_TRY(bsk::launch_hist_row(h->d_obs, h->d_reward, h->d_reason, h->ostride, h->n,
                                         d_obs_hist ? d_obs_hist + (size_t)t * 5 * h->n : nullptr, d_reward_hist ? d_reward_hist + (size_t)t * h->n : nullptr,
                                         d_reason_hist ? d_reason_hist + (size_t)t * h->n : nullptr, h->stream));
        }
        return BSK_OK;
    }
    bsk::StepBuffers b;
    (void)note_capture(h);
    fill_buffers(h, b, nullptr, substeps, 1, false);
    bsk::RolloutBuffers r;
    r.actions = d_actions; r.obs_hist = d_obs_hist; r.reward_hist = d_reward_hist; r.reason_hist = d_reason_hist;
    r.n_steps = n_steps; r.const_action = constant_action;
    hipEvent_t e0, e1;
    { int rc = stamp_events(h, e0, e1); if (rc) return rc; }
    h->last_pair = h->last_tri = false;
    h->last_rollout = true;
    h->last_rollout_act = d_actions != nullptr;
    HIP_TRY(bsk::launch_rollout(h->cfg.gravity_model, h->cfg.n_rw, h->diag, h->sp, b, r, h->block, h->stream, e0, e1));
    h->stats_fresh = false;
    h->wave_sums_fresh = false;
    h->stepped = true;
    return BSK_OK;
}

int bsk_get_obs(bsk_handle* h, double* obs, double* reward, uint8_t* done, uint8_t* done_reason) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    const size_t row = (size_t)h->n * sizeof(double);
    if (obs && h->n == h->ostride) {
        // a batch that fills its stride: the five observation rows are one contiguous block, and the reward row sits right behind them
        // on the device (one allocation) - one plain copy when the host arrays are laid out the same way, two otherwise
        const bool with_reward = reward == obs + 5 * (size_t)h->n;
        HIP_COPY(hipMemcpyAsync(obs, h->d_obs, (with_reward ? 6 : 5) * row, hipMemcpyDeviceToHost, h->stream));
        if (reward && !with_reward) HIP_COPY(hipMemcpyAsync(reward, h->d_reward, row, hipMemcpyDeviceToHost, h->stream));
    } else {
        if (obs)
            HIP_COPY(hipMemcpy2DAsync(obs, row, h->d_obs, (size_t)h->ostride * sizeof(double), row, 5, hipMemcpyDeviceToHost, h->stream));
        if (reward) HIP_COPY(hipMemcpyAsync(reward, h->d_reward, row, hipMemcpyDeviceToHost, h->stream));
    }
    std::vector<unsigned char> why;
    unsigned char* wp = done_reason;
    if (done && !done_reason) { why.resize(h->n); wp = why.data(); }
    if (wp) HIP_COPY(hipMemcpyAsync(wp, h->d_reason, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    if (done)
        for (int i = 0; i < h->n; ++i) done[i] = wp[i] != 0;
    return BSK_OK;
}

int bsk_get_obs_state(bsk_handle* h, double* obs, double* reward, uint8_t* done_reason, double* state) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    const size_t row = (size_t)h->n * sizeof(double), pitch = (size_t)h->stride * sizeof(double), opitch = (size_t)h->ostride * sizeof(double);
    if (obs) HIP_COPY(hipMemcpy2DAsync(obs, row, h->d_obs, opitch, row, 5, hipMemcpyDeviceToHost, h->stream));
    if (reward) HIP_COPY(hipMemcpyAsync(reward, h->d_reward, row, hipMemcpyDeviceToHost, h->stream));
    if (done_reason) HIP_COPY(hipMemcpyAsync(done_reason, h->d_reason, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    if (state) HIP_COPY(hipMemcpy2DAsync(state, row, h->d_state, pitch, row, h->nf, hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_get_obs_rowmajor(bsk_handle* h, double* obs_n5, double* reward, uint8_t* done_reason) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (!h->d_obs_rm) return fail(BSK_EINVAL, "bsk_get_obs_rowmajor needs BSK_FLAG_OBS_ROWMAJOR");
    DeviceGuard guard(h->device);
    const size_t row = (size_t)h->n * sizeof(double);
    if (obs_n5) HIP_COPY(hipMemcpyAsync(obs_n5, h->d_obs_rm, 5 * row, hipMemcpyDeviceToHost, h->stream));
    if (reward) HIP_COPY(hipMemcpyAsync(reward, h->d_reward, row, hipMemcpyDeviceToHost, h->stream));
    if (done_reason) HIP_COPY(hipMemcpyAsync(done_reason, h->d_reason, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_get_obs_device(bsk_handle* h, double** d_obs, double** d_reward, uint64_t** d_done_mask, uint8_t** d_done_reason,
                       int64_t* stride) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (d_obs) *d_obs = h->d_obs;
    if (d_reward) *d_reward = h->d_reward;
    if (d_done_mask) *d_done_mask = (uint64_t*)h->d_done_mask;
    if (d_done_reason) *d_done_reason = h->d_reason;
    if (stride) *stride = h->ostride;
    return BSK_OK;
}

int bsk_get_stream(bsk_handle* h, void** stream) {
    if (!h || !stream) return fail(BSK_EINVAL, "handle/stream is NULL");
    *stream = (void*)h->stream;
    return BSK_OK;
}

int bsk_get_terminal_obs_device(bsk_handle* h, double** d_term_obs, int32_t** d_episodes) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (d_term_obs) *d_term_obs = h->d_term_obs;
    if (d_episodes) *d_episodes = h->d_episodes;
    return BSK_OK;
}

int bsk_get_state_device(bsk_handle* h, double** d_state, int64_t* stride) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (d_state) *d_state = h->d_state;
    if (stride) *stride = h->stride;
    return BSK_OK;
}

int bsk_get_batch_stats(bsk_handle* h, double* reward_sum, int64_t* n_done) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    { int rc = snapshot_stats(h); if (rc) return rc; }
    double s = 0;
    long long d = 0;
    HIP_COPY(hipMemcpyAsync(&s, h->d_stat_sum, sizeof s, hipMemcpyDeviceToHost, h->stream));
    HIP_COPY(hipMemcpyAsync(&d, h->d_stat_done, sizeof d, hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    if (reward_sum) *reward_sum = s;
    if (n_done) *n_done = d;
    return BSK_OK;
}

int bsk_get_state(bsk_handle* h, double* state) {
    if (!h || !state) return fail(BSK_EINVAL, "handle/state is NULL");
    DeviceGuard guard(h->device);
    const size_t row = (size_t)h->n * sizeof(double);
    HIP_COPY(hipMemcpy2DAsync(state, row, h->d_state, (size_t)h->stride * sizeof(double), row, h->nf, hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_set_state(bsk_handle* h, const double* state) {
    if (!h || !state) return fail(BSK_EINVAL, "handle/state is NULL");
    DeviceGuard guard(h->device);
    h->charge_pos = false;       // (the observation buffers no longer describe this state: the kernel reads the charge again)
    const size_t row = (size_t)h->n * sizeof(double);
    HIP_COPY(hipMemcpy2DAsync(h->d_state, (size_t)h->stride * sizeof(double), state, row, row, h->nf, hipMemcpyHostToDevice, h->stream));
    HIP_SYNC(hipStreamSynchronize(h->stream));
    return BSK_OK;
}

int bsk_get_counters(bsk_handle* h, int32_t* steps, int32_t* ticks) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    std::vector<int2> tmp(h->n);
    HIP_COPY(hipMemcpyAsync(tmp.data(), h->d_cnt, (size_t)h->n * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
    HIP_SYNC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < h->n; ++i) {
        if (steps) steps[i] = tmp[i].x & 0xFFFFF;  // high bits carry the FSW phase
        if (ticks) ticks[i] = tmp[i].y;
    }
    return BSK_OK;
}

int bsk_set_counters(bsk_handle* h, const int32_t* steps, const int32_t* ticks) {
    if (!h || !steps || !ticks) return fail(BSK_EINVAL, "handle/steps/ticks is NULL");
    DeviceGuard guard(h->device);
    std::vector<int2> tmp(h->n);
    for (int i = 0; i < h->n; ++i) {
        if (steps[i] < 0 || steps[i] > 0xFFFFF || ticks[i] < 0) return fail(BSK_EINVAL, "steps must be in 0..2^20-1 and ticks >= 0");
        tmp[i].x = steps[i] | ((ticks[i] % h->cfg.fsw_every) << 20);
        tmp[i].y = ticks[i];
    }
    HIP_COPY(hipMemcpyAsync(h->d_cnt, tmp.data(), (size_t)h->n * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    HIP_SYNC(hipStreamSynchronize(h->stream));
    return BSK_OK;
}

int bsk_set_ic_pool(bsk_handle* h, int n_pool, const double* ic_pool) {
    if (!h || !ic_pool) return fail(BSK_EINVAL, "handle/ic_pool is NULL");
    if (!(h->cfg.flags & BSK_FLAG_AUTO_RESET)) return fail(BSK_EINVAL, "handle was not created with BSK_FLAG_AUTO_RESET");
    if (n_pool < 1) return fail(BSK_EINVAL, "n_pool must be >= 1");
    DeviceGuard guard(h->device);
    HIP_SYNC(hipStreamSynchronize(h->stream));
    int rc = ensure_pool_buffers(h, n_pool);
    if (rc) return rc;
    HIP_COPY(hipMemcpy(h->d_pool, ic_pool, (size_t)h->nf * n_pool * sizeof(double), hipMemcpyHostToDevice));
    h->n_pool = n_pool;
    h->pool_charge_pos = true;
    for (int k = 0; k < n_pool; ++k) h->pool_charge_pos = h->pool_charge_pos && ic_pool[(size_t)(BSK_NF_BASE + h->cfg.n_rw + BSK_T_CHARGE) * n_pool + k] > 0.0;
    return BSK_OK;
}

static int ensure_pool_buffers(bsk_handle* h, int n_pool) {
    if (h->d_pool && h->pool_cap < n_pool) { (void)hipFree(h->d_pool); h->d_pool = nullptr; h->n_pool = 0; h->pool_cap = 0; }
    if (!h->d_pool) {
        HIP_TRY(hipMalloc(&h->d_pool, (size_t)h->nf * n_pool * sizeof(double)));
        h->pool_cap = n_pool;
    }
    if (!h->d_term_obs) {
        HIP_TRY(hipMalloc(&h->d_term_obs, (size_t)5 * h->ostride * sizeof(double)));
        HIP_TRY(hipMemset(h->d_term_obs, 0, (size_t)5 * h->ostride * sizeof(double)));
        HIP_TRY(hipMalloc(&h->d_episodes, (size_t)h->ostride * sizeof(int)));
        HIP_TRY(hipMemset(h->d_episodes, 0, (size_t)h->ostride * sizeof(int)));
    }
    return BSK_OK;
}

int bsk_sample_ic_pool(bsk_handle* h, int n_pool, uint64_t seed) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (!(h->cfg.flags & BSK_FLAG_AUTO_RESET)) return fail(BSK_EINVAL, "handle was not created with BSK_FLAG_AUTO_RESET");
    if (n_pool < 1) return fail(BSK_EINVAL, "n_pool must be >= 1");
    DeviceGuard guard(h->device);
    HIP_SYNC(hipStreamSynchronize(h->stream));
    int rc = ensure_pool_buffers(h, n_pool);
    if (rc) return rc;
    HIP_TRY(bsk::launch_sample_pool(h->d_pool, n_pool, h->cfg.n_rw, (unsigned long long)seed, h->cfg.mu, h->stream));
    HIP_SYNC(hipStreamSynchronize(h->stream));
    h->n_pool = n_pool;
    h->pool_charge_pos = true;       // the sampler draws U(8, 20) W h
    return BSK_OK;
}

int bsk_reset_from_pool(bsk_handle* h, const uint8_t* mask) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (h->n_pool == 0) return fail(BSK_EINVAL, "no IC pool staged (bsk_set_ic_pool / bsk_sample_ic_pool)");
    DeviceGuard guard(h->device);
    if (h->stepped) { int rc = snapshot_stats(h); if (rc) return rc; }   // the last step's batch scalars, before its rewards are overwritten
    unsigned char* d_mask = nullptr;
    if (mask) {
        if (!h->d_mask_stage) HIP_TRY(hipMalloc(&h->d_mask_stage, (size_t)h->ostride));   // kept for the handle's lifetime
        d_mask = h->d_mask_stage;
        HIP_COPY(hipMemcpyAsync(d_mask, mask, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(bsk::launch_reset_from_pool(h->d_state, h->stride, h->nf, h->d_pool, h->n_pool, d_mask, h->n, h->d_cnt,
                                        h->d_episodes, h->env_base, reset_out(h), h->stream));
    h->charge_pos = (mask ? h->charge_pos : true) && h->pool_charge_pos;
    { int rc = seal_stats(h); if (rc) return rc; }
    HIP_SYNC(hipStreamSynchronize(h->stream));
    return BSK_OK;
}

int bsk_reset_from_pool_device(bsk_handle* h, const uint8_t* d_mask) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (h->n_pool == 0) return fail(BSK_EINVAL, "no IC pool staged (bsk_set_ic_pool / bsk_sample_ic_pool)");
    DeviceGuard guard(h->device);
    if (h->stepped) { int rc = snapshot_stats(h); if (rc) return rc; }   // the last step's batch scalars, before its rewards are overwritten
    HIP_TRY(bsk::launch_reset_from_pool(h->d_state, h->stride, h->nf, h->d_pool, h->n_pool, d_mask, h->n, h->d_cnt,
                                        h->d_episodes, h->env_base, reset_out(h), h->stream));
    h->charge_pos = (d_mask ? h->charge_pos : true) && h->pool_charge_pos;
    { int rc = seal_stats(h); if (rc) return rc; }
    return BSK_OK;       // asynchronous on the handle's stream: no host data, no copy, no synchronisation
}

int bsk_get_episode_device(bsk_handle* h, double** d_ep_return, double** d_term_return, int32_t** d_term_len, uint8_t** d_done,
                           double** d_obs_rowmajor) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (d_ep_return) *d_ep_return = h->d_ep_return;
    if (d_term_return) *d_term_return = h->d_term_return;
    if (d_term_len) *d_term_len = h->d_term_len;
    if (d_done) *d_done = h->d_done;
    if (d_obs_rowmajor) *d_obs_rowmajor = h->d_obs_rm;
    return BSK_OK;
}

int bsk_set_step_stats(bsk_handle* h, int on) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    h->step_stats = on != 0;
    return BSK_OK;
}

int bsk_get_batch_stats_device(bsk_handle* h, double** d_stats2) {
    if (!h || !d_stats2) return fail(BSK_EINVAL, "handle/d_stats2 is NULL");
    DeviceGuard guard(h->device);
    { int rc = snapshot_stats(h); if (rc) return rc; }
    *d_stats2 = h->d_stats2;
    return BSK_OK;       // asynchronous: the two doubles are valid once the handle's stream has reached this point
}

int bsk_debug_words(bsk_handle* h, uint64_t* words) {
    if (!h || !words) return fail(BSK_EINVAL, "handle/words is NULL");
    DeviceGuard guard(h->device);
    HIP_COPY(hipMemcpyAsync(words, h->d_dbg, (size_t)((h->n + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_debug_counters(int64_t* n_copies, int64_t* n_syncs) {
    if (n_copies) *n_copies = g_n_copies.load(std::memory_order_relaxed);
    if (n_syncs) *n_syncs = g_n_syncs.load(std::memory_order_relaxed);
    return BSK_OK;
}

int bsk_get_ic_pool(bsk_handle* h, double* ic_pool) {
    if (!h || !ic_pool) return fail(BSK_EINVAL, "handle/ic_pool is NULL");
    if (h->n_pool == 0) return fail(BSK_EINVAL, "no IC pool staged (bsk_set_ic_pool / bsk_sample_ic_pool)");
    DeviceGuard guard(h->device);
    HIP_SYNC(hipStreamSynchronize(h->stream));
    HIP_COPY(hipMemcpy(ic_pool, h->d_pool, (size_t)h->nf * h->n_pool * sizeof(double), hipMemcpyDeviceToHost));
    return BSK_OK;
}

int bsk_get_terminal_obs(bsk_handle* h, double* term_obs, int32_t* episodes) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (!h->d_term_obs) return fail(BSK_EINVAL, "no IC pool staged (bsk_set_ic_pool)");
    DeviceGuard guard(h->device);
    const size_t row = (size_t)h->n * sizeof(double);
    if (term_obs)
        HIP_COPY(hipMemcpy2DAsync(term_obs, row, h->d_term_obs, (size_t)h->ostride * sizeof(double), row, 5, hipMemcpyDeviceToHost, h->stream));
    if (episodes) HIP_COPY(hipMemcpyAsync(episodes, h->d_episodes, (size_t)h->n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_set_sim_time(bsk_handle* h, double t) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    h->sim_time = t;
    for (int i = 0; i < 3; ++i) h->sp.pc.sun_r0[i] = h->cfg.sun_r0[i] + h->cfg.sun_v[i] * t;
    return BSK_OK;
}

int bsk_set_env_base(bsk_handle* h, int64_t env_base) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    if (env_base < 0 || env_base > 0xFFFFFFFFll) return fail(BSK_EINVAL, "env_base must be in 0..2^32-1");
    h->env_base = (unsigned)env_base;
    return BSK_OK;
}

int bsk_sync(bsk_handle* h) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    SYNC_CHECKED(h);
    return BSK_OK;
}

int bsk_profile_begin(bsk_handle* h, int capacity) {
    if (!h || capacity < 1) return fail(BSK_EINVAL, "handle is NULL or capacity < 1");
    DeviceGuard guard(h->device);
    while ((int)h->ev.size() < 2 * capacity) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    h->ev_used = 0;
    h->ev_seq = 0;
    h->prof = true;
    return BSK_OK;
}

int bsk_profile_set_stride(bsk_handle* h, int stride) {
    if (!h || stride < 1) return fail(BSK_EINVAL, "handle is NULL or stride < 1");
    h->ev_stride = stride;
    return BSK_OK;
}

int bsk_profile_end_samples(bsk_handle* h, double* mean_kernel_ms, int* n_launches, float* samples_ms, int cap) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    h->prof = false;
    HIP_SYNC(hipStreamSynchronize(h->stream));
    double tot = 0.0;
    const int n = h->ev_used / 2;
    for (int k = 0; k < n; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * k], h->ev[2 * k + 1]));
        tot += ms;
        if (samples_ms && k < cap) samples_ms[k] = ms;
    }
    if (mean_kernel_ms) *mean_kernel_ms = n ? tot / n : 0.0;
    if (n_launches) *n_launches = n;
    h->ev_used = 0;
    return BSK_OK;
}

int bsk_profile_end(bsk_handle* h, double* mean_kernel_ms, int* n_launches) {
    return bsk_profile_end_samples(h, mean_kernel_ms, n_launches, nullptr, 0);
}

int bsk_kernel_info(bsk_handle* h, char* name, int name_cap, int* vgprs, int* lds_bytes, int* block, int* grid) {
    if (!h) return fail(BSK_EINVAL, "handle is NULL");
    DeviceGuard guard(h->device);
    const bool sh = h->cfg.gravity_model == BSK_GRAV_SH;
    // (the kernel of the LAST launch: the pair form is chosen per launch by its number of sub-steps)
    const void* fp = h->last_rollout ? bsk::rollout_kernel_ptr(h->cfg.gravity_model, h->cfg.n_rw, h->diag, h->last_rollout_act)
                                     : bsk::step_kernel_ptr(h->cfg.gravity_model, h->cfg.n_rw, h->diag, h->sp.feat, h->sp.sh_form, h->last_pair, h->last_tri);
    if (!fp) return fail(BSK_EINVAL, "no kernel variant for this config");
    hipFuncAttributes at;
    HIP_TRY(hipFuncGetAttributes(&at, fp));
    if (name && name_cap > 0 && h->last_rollout)
        std::snprintf(name, name_cap, "rollout_kernel<%s,%d,%s,%s>", h->cfg.gravity_model == BSK_GRAV_PM ? "PM" : "PM_J2", h->cfg.n_rw, h->diag ? "diag" : "full",
                      h->last_rollout_act ? "actions" : "constant");
    else if (name && name_cap > 0)
        std::snprintf(name, name_cap, "step_kernel<%s,%d,%s>",
                      h->cfg.gravity_model == BSK_GRAV_PM ? "PM" : (h->cfg.gravity_model == BSK_GRAV_PM_J2 ? "PM_J2" : (h->sp.sh_form == 5 ? "SH/dpp2" : (h->sp.sh_form == 4 ? "SH/dpp" : "SH/scalar"))), h->cfg.n_rw,
                      h->sp.feat >= 2 ? (h->sp.feat == 3 ? (h->diag ? "diag,scenario/generic-facets" : "full,scenario/generic-facets")
                                                         : (h->last_tri ? "diag,scenario,tri" : (h->last_pair ? "diag,scenario,pair" : (h->diag ? "diag,scenario" : "full,scenario"))))
                                      : (h->sp.feat == 1 ? (h->last_pair ? "diag,power,pair" : (h->diag ? "diag,power" : "full,power"))
                                                         : (h->sp.feat == -1 ? (h->diag ? "diag,lds-scratch" : "full,lds-scratch") : (h->diag ? "diag" : "full"))));
    if (vgprs) *vgprs = at.numRegs;
    if (lds_bytes) *lds_bytes = (int)at.sharedSizeBytes;
    // the two-wave harmonics form launches 256-thread workgroups of 2 x 64 spacecraft x 2 halves
    const int blk = (sh && h->sp.sh_form == 5) ? 256 : (h->last_tri ? 192 : (h->last_pair ? 128 : h->block));
    if (block) *block = blk;
    if (grid) *grid = (sh && h->sp.sh_form == 5) ? (h->n + 127) / 128 : ((h->last_pair || h->last_tri) ? (h->n + 63) / 64 : (h->n + blk - 1) / blk);
    return BSK_OK;
}

}  // extern "C"
