// bsk_device.hpp — per-spacecraft fp64 physics for gfx950 (CDNA4), one spacecraft per lane.
//
// The step kernel is fp64-VALU-issue bound (measured: one wave per SIMD keeps the VALU active 91 %
// of its cycles at 4.1 cycles per instruction), so this file is written for minimum VALU
// instruction count:
//   * constants the RK4 loop needs (HotCfg) travel BY VALUE in the kernarg segment and stay in
//     SGPRs; HotCfg is specialised on <NRW, DIAG> and the wheel geometry is parked in VGPRs so that
//     the loop fits the 102-SGPR budget (an overflow turns into v_readlane/v_writelane spill
//     traffic on the VALU, the bottleneck);
//   * constants only the 1 Hz FSW chain needs (ColdCfg) sit behind a pointer and are loaded where
//     used, so they hold no SGPRs across the loop;
//   * the wheels are integrated through their total momentum (closed ODE) and recovered exactly at
//     the end of the step (rk4_step);
//   * 1/|r| is one v_rsq_f64 + one cubic Newton step (6 ops) instead of IEEE sqrt + IEEE divide;
//   * cross products are folded into FMA chains; wheel friction is branch-free.
//
// What this replaces: the Basilisk modules wired by the reference scenario
//   spacecraftPlus + RK4        basilisk_env/simulators/leoPowerAttitudeSimulator.py:213-259,356
//   gravityEffector             :217-232   (J2 / harmonics: opNav_models/BSK_OpNavDynamics.py:211-214)
//   reactionWheelStateEffector  :301-310 + dynamics/effectorPrimatives/actuatorPrimatives.py:7-63
//   extForceTorque              :291-298
//   hillPoint / inertial3D / attTrackingError / MRP_Feedback / rwMotorTorque   :407-449, 481-486
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/bskgpu.h"
#include "bsk_probes.hpp"

namespace bsk {

// A taken branch costs a wave that runs alone on its SIMD ~55 clocks of instruction fetch (SQ_WAIT_ANY of the
// full-scenario level: 12 taken branches per tick = 15 % of its cycles).  Blocks that are rarely entered are marked so
// that the compiler lays them out of line and the common path falls through.
#define BSK_LIKELY(x) __builtin_expect(!!(x), 1)
#define BSK_UNLIKELY(x) __builtin_expect(!!(x), 0)

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x))};
}
// c + a x b and c - a x b, folded into FMA chains (6 ops)
__device__ __forceinline__ V3 add_cross(V3 c, V3 a, V3 b) {
    return V3{fma(a.y, b.z, fma(-a.z, b.y, c.x)), fma(a.z, b.x, fma(-a.x, b.z, c.y)), fma(a.x, b.y, fma(-a.y, b.x, c.z))};
}
__device__ __forceinline__ V3 sub_cross(V3 c, V3 a, V3 b) {
    return V3{fma(-a.y, b.z, fma(a.z, b.y, c.x)), fma(-a.z, b.x, fma(a.x, b.z, c.y)), fma(-a.x, b.y, fma(a.y, b.x, c.z))};
}
// y = a*x + y
__device__ __forceinline__ V3 axpy(double a, V3 x, V3 y) { return V3{fma(a, x.x, y.x), fma(a, x.y, y.y), fma(a, x.z, y.z)}; }
__device__ __forceinline__ V3 mv(const double* m, V3 v) {
    return V3{fma(m[0], v.x, fma(m[1], v.y, m[2] * v.z)), fma(m[3], v.x, fma(m[4], v.y, m[5] * v.z)),
              fma(m[6], v.x, fma(m[7], v.y, m[8] * v.z))};
}

// 1/sqrt(x) for normal positive x: hardware estimate (v_rsq_f64, ~2^-23 relative) + one cubic
// Newton step y (1 + e/2 + 3e^2/8), e = 1 - x y^2  ->  error ~e^3, i.e. rounding-limited (~1 ulp).
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double t = x * y;
    double e = fma(-t, y, 1.0);
    double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
}

// 1/x for normal x: v_rcp_f64 estimate + two Newton steps (error ~1 ulp), 5 ops instead of the
// ~12-instruction IEEE divide expansion.
__device__ __forceinline__ double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
// sqrt(x) for x >= 0 (0 -> 0): x * rsqrt(x)
__device__ __forceinline__ double sqrt_nr(double x) { return x > 0.0 ? x * rsqrt_nr(x) : 0.0; }

// Tell the compiler a value is wave-uniform (keeps it in SGPRs so that loads/stores through it
// take the saddr form).  Free when the value already lives in SGPRs.
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// uniform pointer into GLOBAL memory (address space 1 is spelled out because an integer->pointer
// round trip would otherwise fall back to flat addressing)
template <class T>
using gptr = T __attribute__((address_space(1)))*;
template <class T>
__device__ __forceinline__ gptr<T> uniform_ptr(T* p) {
    return (gptr<T>)p;   // explicit flat -> global cast; the value itself comes from a scalar load
}

// Coalesced SoA access: uniform field base (SGPR pair) + one 32-bit per-lane byte offset shared by
// every field -> global_load/store with the saddr form, no 64-bit VGPR address arithmetic.
__device__ __forceinline__ double ldf(const double* __restrict__ base, uint32_t boff) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff);
}
// (stores: the zero-extension of the offset has to be visible in the store's own basic block for the saddr form to be
// selected; left to itself it is computed once at the top of the kernel, every store then adds a 64-bit register pair to
// its base - one VALU instruction per store - and takes the vaddr form.  The empty asm makes each use's offset a value of
// its own.)
__device__ __forceinline__ void stf(gptr<double> base, uint32_t boff, double v) {
    asm("" : "+v"(boff));
    *(gptr<double>)((gptr<char>)base + boff) = v;
}

// --------------------------------------------------------------------------------------------
// Hot constants (SGPR-resident).  M3 = 3 (diagonal) or 9 (full, row-major) doubles per matrix.
template <int NRW, bool DIAG>
struct HotCfg {
    double h, h2, h3, h6;  // dt, dt/2, dt/3, dt/6
    double nmu, j2k;       // -mu,  1.5 * J2 * mu * req^2
    double Dm[DIAG ? 3 : 9];  // I_sc - sum Js g g^T   (the hub's inertia without the wheels' spin-axis part)
    double Di[DIAG ? 3 : 9];  // its inverse
    double W[DIAG ? 3 : 9];   // sum Js g g^T  (wheel momentum as a function of omega, see rk4_step)
    double g[NRW > 0 ? NRW : 1][3];
    double js[NRW > 0 ? NRW : 1], ijs[NRW > 0 ? NRW : 1];
    double fc;
    int32_t fsw_every, sh_degree;
    // harmonics walk (bsk_capi.hip: bsk_set_gravity_sh): first column of the second half; bodies of the
    // whole padded stream; bodies of each half and first chunk of the second half (two-wave form)
    int32_t sh_split, sh_bodies;
    int32_t sh_bodies0, sh_bodies1, sh_chunk1, pad_;
    // spherical harmonics (GRAV == BSK_GRAV_SH only; unused kernarg fields cost no SGPRs)
    const double* sh_tab;   // fused Pines stream, 8 doubles per (l, m) step, iteration order
    double mu_over_req, req, inv_req, planet_rate;
};

// Cold constants of the 1 Hz control law (device memory, s_load-ed inside the FSW block only).
struct ColdCfg {
    double inertia[9];
    double map[BSK_MAX_RW][3];  // rwMotorTorque pseudo-inverse rows
    double u_max, u_min;
    double K, P;
    double sigma_R0N[3];
    // facetDragDynamicEffector geometry (read only inside the drag branch)
    double facet_acd[8];        // area * Cd
    double facet_n[8][3], facet_r[8][3];
    // facets whose normals are +-body axes (all eight of the reference's): per axis k, half sum [0] and
    // half difference [1] of the +e_k and -e_k facets' area*Cd and area*Cd*r (facet_axis = 1 when every
    // facet is of that kind, 2 when in addition every fa_r[.][k][j != k] is exactly zero)
    double fa_c[2][3];
    double fa_r[2][3][3];
    int32_t n_facets, n_thr;
    int32_t facet_axis, pad2_;
    // desaturation chain (FEAT_FULL + BSK_FLAG_DESAT)
    double thr_map[BSK_MAX_THR][3];   // thrForceMapping: [D]^T ([D][D]^T)^-1, D_i = r_i x dir_i
    double thr_f[BSK_MAX_THR][3], thr_l[BSK_MAX_THR][3];  // force / torque of thruster i at full thrust, body frame
    double hs_min, inv_max_thrust, thr_min_fire_time, thr_min_on_time;
    double gs[BSK_MAX_RW][3], js[BSK_MAX_RW];
    int32_t thr_max_counter;
    int32_t fsw_lag;   // bsk_config.fsw_lag: MRP_Feedback consumes the previous FSW tick's att_guidance
    int32_t nav_lag, pad3_;   // bsk_config.nav_lag: FSW ticks run before the dynamics task of their time
    // wave-uniform constants of the full-scenario kernels: three rows of 16 doubles, fetched lane-wise into three
    // VGPR pairs (lane l holds entry l & 15 of each row) and fed to the FMAs through the DPP row broadcast (KTab)
    double kt[80];
    // force / torque sums of every subset of the thrusters at full thrust, body frame: row m = sum over the bits
    // of m of (thr_f[i], thr_l[i]) added in ascending i (the oracle's order), so an 8-term conditional sum with
    // 48 table loads per integrator stage becomes one 6-double row picked by the activity mask
    double thr_tab[1 << BSK_MAX_THR][6];
};
// entries of the broadcast table (ColdCfg::kt): row A, row B, row C
enum { KA_G = 0, KA_JS = 12 };                                 // wheel spin axes g[i][k] at 3 i + k, Js_i
enum { KB_IJS = 0, KB_FAC = 4, KB_FAD = 10 };                  // 1/Js_i, facet half sums / differences (6 + 6)
enum { KC_IMASS = 0, KC_NB = 1, KC_KFLUX = 4, KC_RHO0 = 5, KC_NIH = 6, KC_REQIH = 7, KC_RSKIP = 8, KC_LOG2E = 9, KC_I6 = 10, KC_I24 = 11, KC_I120 = 12, KC_I720 = 13 };   // 1/m, panel normal, ...
enum { KD_JG = 0, KD_HIJS = 12 };                              // row D: Js_i g[i][k] at 3 i + k, dt / Js_i
enum { KE_POLY = 0, KE_NLN2HI = 14, KE_NLN2LO = 15 };          // row E: rho0 / k!, k = 0..13 (atmosphere_density); -ln2 in two parts

// Guidance / observation / reward constants: by value in the kernarg (used once per launch, outside
// the RK4 loop, so they may be parked in VGPR lanes across it at no cost to the loop).
struct ObsCfg {
    double sigma_R0N[3];
    double inv_wheel_limit, charge_scale, reward_mult, failure_penalty, r_min2;
    int32_t max_length, pad_;
};

// Wheel geometry lives in VGPRs (same value in every lane): g, js, 1/js are 20 doubles for four
// wheels, and together with the rest of HotCfg they overflow the SGPR file; a VGPR operand costs
// a VALU instruction nothing, an SGPR spill costs a v_readlane per use.
__device__ __forceinline__ double to_vgpr(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
template <int NRW>
struct WheelV {
    double g_[NRW > 0 ? NRW : 1][3];
    double js_[NRW > 0 ? NRW : 1], ijs_[NRW > 0 ? NRW : 1];
    template <class Hot>
    __device__ __forceinline__ void load(const Hot& c) {
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            g_[i][0] = to_vgpr(c.g[i][0]); g_[i][1] = to_vgpr(c.g[i][1]); g_[i][2] = to_vgpr(c.g[i][2]);
            js_[i] = to_vgpr(c.js[i]); ijs_[i] = to_vgpr(c.ijs[i]);
        }
    }
    // head of an RK4 step: T = sum tq_i g_i, p += sum (Js_i Om_i) g_i (the caller's W w0, see rk4_step), tqj_i = tq_i / Js_i
    __device__ __forceinline__ void head(const double* tq, const double* Om, V3& T, V3& p, double* tqj) const {
        T = mk(0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const V3 g = mk(g_[i][0], g_[i][1], g_[i][2]);
            T = axpy(tq[i], g, T);
            p = axpy(js_[i] * Om[i], g, p);
            tqj[i] = tq[i] * ijs_[i];
        }
    }
    // wheel speeds at the end of the step without the hub's reaction: Om_i + h tq_i / Js_i
    __device__ __forceinline__ void bases(double h, const double* tq, const double* tqj, const double* Om, double* base) const {
#pragma unroll
        for (int i = 0; i < NRW; ++i) base[i] = fma(h, tqj[i], Om[i]);
    }
    // tail: Om_i = base_i - g_i . dw   (accumulated z, y, x: the order of the fma chain below)
    __device__ __forceinline__ void tail(V3 dw, const double* base, double* Om) const {
#pragma unroll
        for (int i = 0; i < NRW; ++i) Om[i] = fma(-g_[i][0], dw.x, fma(-g_[i][1], dw.y, fma(-g_[i][2], dw.z, base[i])));
    }
};

// Feature level of a kernel variant (template parameter FEAT):
//   0 bare propagator (the bench headline), 1 + power system, 2 full scenario = power + the
//   wave-uniform runtime switches below (Sun third-body gravity, atmospheric drag).
//   3 = level 2 for facet sets that are not "axis-aligned with every centre on its own normal axis" (the
//   reference's eight are): tables read at each use / loop over the facet list.  Kept out of level 2 so that the
//   kernel the drop-in env runs carries neither their code nor their registers.
//  -1 = level 0 with the RK4 accumulator staged in LDS (BSK_FLAG_LDS_SCRATCH; the north star's "per-spacecraft
//   RK4 scratch staged in LDS"): 15 doubles per lane leave the register file between the stages.
enum { FEAT_LDSS = -1, FEAT_BARE = 0, FEAT_POWER = 1, FEAT_FULL = 2, FEAT_FULLG = 3 };
template <int FEAT>
constexpr bool is_full() { return FEAT == FEAT_FULL || FEAT == FEAT_FULLG; }

// Sun third-body gravity (leoPowerAttitudeSimulator.py:227-232) and exponentialAtmosphere +
// facet drag (:265-284, parameters :146-148); FEAT_FULL only.
struct ExtraCfg {
    double mu_sun;             // 0 = third body off
    double base_density;       // 0 = drag off
    double inv_scale_height, inv_mass, rho_skip;
    int32_t desat, pad_;       // BSK_FLAG_DESAT
};

// --------------------------------------------------------------------------------------------
// Power system: eclipse -> simpleSolarPanel -> simpleBattery <- simplePowerSink,
// reference wiring leoPowerAttitudeSimulator.py:286-288, 326-345, 363-366, parameters :158-167.
struct PowerCfg {
    double nB[3];            // panel normal, body frame
    double kflux;            // panel_area * efficiency * solar_flux(1 AU) * AU^2   [W m^2]
    double draw, cap;        // sink power [W] (negative), battery capacity [W s]
    double req, rs_plus, rs_minus;  // planet radius, R_sun + R_planet, R_sun - R_planet
    double rsun;
    double sun_r0[3], sun_v[3];
};

// Per-launch Sun geometry (the Sun is held over the env step like the 180 s SPICE task): everything
// that depends only on the Sun position is computed once, so the per-step eclipse test needs no
// transcendental function outside the penumbra.
struct SunGeom {
    V3 sun;
    double ism;              // 1 / |sun|
    double re_sf1, re_sf2;   // Re / sin f1, Re / sin f2   (penumbra / umbra cone half-angles)
    double tf1, tf2;         // tan f1, tan f2
};

__device__ __forceinline__ SunGeom sun_setup(const PowerCfg& pc, double t0) {
    SunGeom g;
    g.sun = mk(fma(pc.sun_v[0], t0, pc.sun_r0[0]), fma(pc.sun_v[1], t0, pc.sun_r0[1]), fma(pc.sun_v[2], t0, pc.sun_r0[2]));
    g.ism = rsqrt_nr(dot(g.sun, g.sun));
    const double sf1 = pc.rs_plus * g.ism, sf2 = pc.rs_minus * g.ism;
    g.re_sf1 = pc.req * rcp_nr(sf1);
    g.re_sf2 = pc.req * rcp_nr(sf2);
    g.tf1 = sf1 * rsqrt_nr(fma(-sf1, sf1, 1.0));
    g.tf2 = sf2 * rsqrt_nr(fma(-sf2, sf2, 1.0));
    return g;
}

// asin(s) for |s| <= 1/32 (series to s^9: truncation < 1e-18 relative)
__device__ __forceinline__ double asin_small(double s) {
    const double z = s * s;
    return s * fma(z, fma(z, fma(z, fma(z, 35.0 / 1152.0, 15.0 / 336.0), 3.0 / 40.0), 1.0 / 6.0), 1.0);
}

// asin(sqrt z) / sqrt z on z in [0, 1/4]: degree-12 polynomial (Chebyshev fit in 50-digit arithmetic, maximum
// error 3e-17).  The two inverse functions the penumbra needs are built on it with half-angle reductions,
// branch-free: the library's asin / atan2 cost the wave ~400 instructions with all their argument-range
// branches, and the wave pays them whenever one lane is in the penumbra.
__device__ __forceinline__ double asin_core(double z) {
    double p = 0.03187962140081284;
    p = fma(p, z, -0.016187392271599134);
    p = fma(p, z, 0.019513468251252167);
    p = fma(p, z, 0.0065293020047365695);
    p = fma(p, z, 0.012170138592391726);
    p = fma(p, z, 0.01388484282640208);
    p = fma(p, z, 0.01735977964134998);
    p = fma(p, z, 0.022371749733164054);
    p = fma(p, z, 0.03038195969768514);
    p = fma(p, z, 0.044642856805998936);
    p = fma(p, z, 0.07500000000385201);
    p = fma(p, z, 0.16666666666664942);
    return fma(p, z, 1.0);
}
// asin(s) for s in [0, 1]
__device__ __forceinline__ double asin01(double s) {
    const bool big = s > 0.5;
    const double z = big ? fma(-0.5, s, 0.5) : s * s;          // sin^2 of half the complementary angle | s^2
    const double q = big ? sqrt_nr(z) : s;
    const double r = q * asin_core(z);
    return big ? fma(-2.0, r, 1.57079632679489661923) : r;
}
// the angle theta in [0, pi] with cos theta = x/a, sin theta = y/a (y >= 0, x^2 + y^2 = a^2): near the axes the
// half angle comes from the SINE (y), so the result is well conditioned everywhere (acos(x/a) is not)
__device__ __forceinline__ double angle_xy(double x, double y, double a) {
    const double ia = rcp_nr(a), t = x * ia, sy = y * ia, at = fabs(t);
    const bool big = at > 0.5;
    const double z = big ? 0.5 * sy * sy * rcp_nr(1.0 + at) : t * t;
    const double q = big ? sqrt_nr(z) : at;
    const double r = q * asin_core(z);
    const double PI = 3.14159265358979323846;
    const double base = big ? (t > 0.0 ? 0.0 : PI) : 0.5 * PI;
    const double k = big ? (t > 0.0 ? 2.0 : -2.0) : (t > 0.0 ? -1.0 : 1.0);
    return fma(k, r, base);
}

// The reference's lens-area formula as published, for the geometries the fast path below does not cover (solar
// disc not small against the planet's: never in LEO).  Kept out of line: its library calls (asin / acos with all
// their range branches) would otherwise set the register allocation of the drain loop that inlines the fast path.
__device__ __attribute__((noinline)) double percent_shadow_generic(double sa, double sb, double cc) {
    const double PI = 3.14159265358979323846;
    const double a = asin(sa), b = asin(sb), c = acos(cc);
    if (c < a - b) return 1.0 - (b * b) / (a * a);                              // annular
    const double x = (c * c + a * a - b * b) / (2.0 * c), y = sqrt(fmax(a * a - x * x, 0.0));
    const double area = a * a * acos(x / a) + b * b * acos((c - x) / b) - c * y;
    return 1.0 - area / (PI * a * a);
}

// visible fraction of the solar disc inside the shadow cones: total eclipse is decided on cosines
// (no inverse trigonometry); only partial / annular phases reach the inverse functions.
// The reference's lens-area formula is  1 - [a^2 acos(x/a) + b^2 acos((c-x)/b) - c y] / (pi a^2),
// x = (c^2 + a^2 - b^2)/2c, y = sqrt(a^2 - x^2), with the apparent radii a = asin(sa), b = asin(sb) and
// the separation c = acos(cc).  When the solar disc is small against the planet's (always in LEO:
// sa ~ 4.7e-3, sb ~ 0.93) the same quantities follow without library calls and without the formula's
// cancellations: a and theta2 = acos((c-x)/b) = asin(y/b) by series, delta = c - b from
// sin(delta) = sin c cos b - cos c sin b by series, b = asin01(sb), theta1 = acos(x/a) = angle_xy(x, y, a),
// and b^2 theta2 - c y = b^2 (theta2 - y/b) - delta y.  (The wave executes this path whenever one of its
// 64 spacecraft is in the penumbra, and with one wave per SIMD the slowest wave sets the kernel time.)
__device__ __forceinline__ double percent_shadow(const PowerCfg& pc, V3 r_HB, V3 r, double r2) {
    const double nh2 = dot(r_HB, r_HB);
    const double inh = rsqrt_nr(nh2), ins = rsqrt_nr(r2);
    const double sa = fmin(pc.rsun * inh, 1.0), sb = fmin(pc.req * ins, 1.0);   // sin a, sin b
    const double cc = fmax(fmin(-dot(r, r_HB) * inh * ins, 1.0), -1.0);         // cos c
    const double ca = sqrt_nr(fma(-sa, sa, 1.0)), cb = sqrt_nr(fma(-sb, sb, 1.0));
    if (sb > sa && cc > fma(cb, ca, sb * sa)) return 0.0;                       // c < b - a : total
    if (cc <= fma(cb, ca, -(sb * sa))) return 1.0;                              // c >= a + b : none
    const double PI = 3.14159265358979323846;
    if (sa < 0.03125 * sb) {
        const double a = asin_small(sa), b = asin01(sb);
        const double sc = sqrt_nr(fma(-cc, cc, 1.0));
        const double d = asin_small(fma(sc, cb, -(cc * sb)));                   // c - b, |d| <= a
        const double c = b + d, a2 = a * a;
        const double x = fma(d, fma(2.0, b, d), a2) * rcp_nr(2.0 * c);
        const double y = sqrt_nr(fmax(fma(-x, x, a2), 0.0));
        const double th1 = angle_xy(x, y, a);
        const double w = y * rcp_nr(b), z = w * w;                              // theta2 - y/b = w^3 (1/6 + ...)
        const double t2 = w * z * fma(z, fma(z, fma(z, 35.0 / 1152.0, 15.0 / 336.0), 3.0 / 40.0), 1.0 / 6.0);
        const double area = fma(a2, th1, fma(b * b, t2, -(d * y)));
        return 1.0 - area * rcp_nr(PI * a2);
    }
    return percent_shadow_generic(sa, sb, cc);
}

// The EnvTask of one dyn tick (eclipse -> simpleSolarPanel -> simpleBattery <- simplePowerSink), split so that the
// one expensive piece — the partially eclipsed disc, percent_shadow, ~250 issue slots — leaves the tick loop.
// With one spacecraft per lane the wave pays that path whenever ANY of its 64 spacecraft is in the penumbra, and
// with one wave per SIMD the kernel lasts as long as its slowest wave: a single spacecraft whose orbit grazes the
// shadow cone kept its wave on the slow path at every tick of the launch.  Instead, each tick only classifies
// (shadow_quick: lit / umbra / partial) and records what the battery update needs:
//   g[t][lane]  the tick's panel power per unit of lit disc,  s[t][lane]  its shadow factor when known,
// and a partially eclipsed tick appends (position, owner lane, slot) to a per-wave queue in LDS.  After at most
// PEN_SLOTS ticks (one FSW period of the reference) the wave drains the queue COOPERATIVELY — entry e is
// evaluated by lane e mod 64, whoever owns it, so ten penumbra ticks of one spacecraft cost one pass of
// percent_shadow instead of ten — and every lane replays its battery updates in tick order (three operations per
// tick).  Same arithmetic on the same inputs as the tick-by-tick form, hence the same results bit for bit.
// Round 3: the record is three FSW periods long.  Measured (profiles/r03/rejected/power_system_forms.txt): the per-tick
// arithmetic is nearly free, what costs is the drain - percent_shadow is ~250 dependent fp64 instructions, ~2 us for a wave
// alone on its SIMD - and with 64 spacecraft per wave some lane is in the penumbra during most 10-tick windows, so
// practically every flush paid it (0.4 ms of the power level's 2.3 ms at K = 1800).  One pass serves up to 64 queue entries
// for the same latency, so the record now spans PEN_SLOTS = 30 ticks (a third of the passes); the queue holds PEN_QCAP
// entries and a tick that finds it full evaluates its factor on the spot (never in LEO: 192 entries are three spacecraft in
// the penumbra for the whole record), so the worst case costs time, not correctness.
constexpr int PEN_CHUNK = 10;                 // RK4 ticks per trip of the inner loop at most (third-body anchor)
constexpr int PEN_SLOTS = 30;                 // ticks recorded between two flushes
constexpr int PEN_QCAP = 192;                 // penumbra queue entries per record
struct PowerLds {                             // one per wavefront, in dynamic LDS (38.5 KB: four waves per CU fit the 160 KB)
    double g[PEN_SLOTS][64];
    double s[PEN_SLOTS][64];
    double sun[3][64];                        // each lane's Sun position of this launch
    double qr[3][PEN_QCAP];                   // queue: spacecraft position
    int qown[PEN_QCAP];                       // queue: owner lane | slot << 8
    int qcount, pad_[3];
};
// LDS pointers are spelled with their address space (through a generic pointer the accesses would be flat)
typedef PowerLds __attribute__((address_space(3))) * LdsP;

// Pair form (SPLIT == 2; round 3): TWO cooperating waves per 64 spacecraft - a dynamics wave (RK4 with gravity, Sun, drag,
// thrusters, wheels: Basilisk's DynamicsProcess) and a flight-software + environment wave (the 1 Hz FSW chain,
// desaturation, eclipse -> panel -> battery: its FSWProcess and EnvTask) - so that 65 536 spacecraft are 2 048 waves, two
// per SIMD, and what was a serial 24 % tail of every env step (measured: dynamics alone 2.87 of 3.53 ms) runs beside the
// integration instead of after it.  They exchange through this block: the dynamics wave leaves (r, sigma) of every tick in a
// ring, double-buffered by chunk of PAIR_CHUNK ticks, and the state at FSW ticks in `box`; the other wave answers in `box`
// with the commands (wheel torques, thruster burst) and, at the end of the launch, with battery charge, shadow factor and
// the FSW bookkeeping.  80 KB per pair with chunks of 10 ticks (38.9 KB with chunks of 4, which four pairs per CU would need).
// = PEN_CHUNK: the same chunks as the single-wave form (same third-body anchors: bit-identical results at every level);
// 80 KB of LDS per pair - the form is used up to one pair per CU
constexpr int PAIR_CHUNK = 10;
struct PairLds {
    double rr[2][3][PAIR_CHUNK][64];          // ring: position after each tick of the chunk
    double rs[2][3][PAIR_CHUNK][64];          // ring: sigma_BN after each tick
    double box[16][64];                       // D -> F: state at an FSW tick;  F -> D: commands, final values (aliased)
    double sfac[PAIR_CHUNK][64];              // shadow factors the cooperative drain filled in
    double sun[3][64];                        // each lane's Sun position of this launch
    double lext[3][64];                       // each lane's disturbance torque (parked here: the dynamics wave's tick loop has
                                              // 256 registers, and what does not fit goes to scratch memory - the first version
                                              // reloaded two spilled doubles per tick and lost 15 % to it)
    int qown[PAIR_CHUNK * 64];                // penumbra queue: owner lane | slot << 8
    int qcount, pad_[3];
};
typedef PairLds __attribute__((address_space(3))) * PairP;

constexpr int PART_ALL = 0, PART_ROT = 1, PART_TRA = 2;   // which half of the spacecraft a wave integrates

// Three-wave form (SPLIT == 3; round 3): the pair form with its dynamics wave cut in two - a translational wave (r, v) and a
// rotational wave (sigma, omega, wheels) of the SAME 64 spacecraft, each on its own SIMD.  For batches that leave SIMDs idle
// (one workgroup per CU and fewer) an env step is a serial chain of 1 800 ticks, and this shortens the chain: the two halves
// are coupled only through the drag / thrust rotation (attitude -> force) and the drag torque (velocity -> torque), three
// doubles each way per integrator stage.  They are exchanged through this block WITHOUT a barrier (the third wave must not
// take part, and a barrier would put a round trip on the critical path of every stage):
//  * the k-th value a wave publishes goes to slot k & 3 - its rows first, then the slot's per-lane tag k + 1; the k-th value
//    the partner consumes is read from the same slot - the tag FIRST, the rows after it, in one batch - and accepted when
//    every lane's tag is k + 1.  LDS operations of one wave execute in order, so a current tag implies current rows;
//  * a value is published as soon as it exists (the next stage's attitude right after the kinematics, long before the stage's
//    torque is known) and its reads are issued a stage's independent work ahead of its first use (`prefetch` / `finish`), so
//    that in the steady state neither wave waits: the partner's value has been sitting in LDS for tens of instructions;
//  * four slots: a wave overwrites slot k & 3 with value k + 4 only after it has consumed the partner's value k + 2 or later,
//    which the partner published after consuming value k (rk4_step_part's order of publish / consume per stage);
//  * a poll that does not see its tag within TRI_SPIN_LIMIT reads gives up for the rest of the launch and raises `err` (the
//    observation is then NaN): every wave reaches the end of the kernel whatever the other one does.
constexpr int TRI_SPIN_LIMIT = 1 << 20;
enum { BSK_DEVERR_TRI_EXCHANGE = 1 };      // values of the handle's device error word (bsk_capi.hip: check_device_error)
struct TriX {
    double v[4][4][64];                       // translational -> rotational: stage velocity; row 3: the next tick's density
    double s[4][3][64];                       // rotational -> translational: stage attitude
    int tv[4][64], ts[4][64];                 // per-lane tags of the slots
    int err, pad0_;
    int pad_[4];                              // (probe builds: two 64-bit words)
};
struct TriLds {
    PairLds p;
    TriX x;
};
typedef TriX __attribute__((address_space(3))) * TriXP;
typedef TriLds __attribute__((address_space(3))) * TriP;

// The exchange's LDS accesses are RELAXED ATOMICS of workgroup scope: the compiler neither merges nor drops them, and - unlike
// volatile accesses, each of which it follows with a wait for its completion - it leaves a batch of reads in flight until
// the first use of a result.  What relaxed atomics do NOT promise is their order against each other (different addresses), and
// the protocol needs two orders: rows BEFORE the tag on the publishing side, tag BEFORE the rows on the consuming side.
//  * against the COMPILER: tri_order() between the two groups - a compiler barrier, no instruction;
//  * in the HARDWARE: the LDS operations of one wave are issued and executed in program order (one in-order queue per CU),
//    so no wait is needed between them.  A release store / acquire load of workgroup scope would say the same thing
//    portably, and costs an s_waitcnt vmcnt(0) lgkmcnt(0) per publish / consume on this target - eight drains of the
//    wave's whole memory pipeline per tick on the critical path of the form whose only purpose is latency.
__device__ __forceinline__ void tri_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void tri_st(double __attribute__((address_space(3))) * p, double v) {
    __hip_atomic_store((long long __attribute__((address_space(3)))*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double tri_ld(const double __attribute__((address_space(3))) * p) {
    return __longlong_as_double(__hip_atomic_load((long long __attribute__((address_space(3)))*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void tri_sti(int __attribute__((address_space(3))) * p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int tri_ldi(int __attribute__((address_space(3))) * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct TriPend {     // a consume in flight
    int tag;
    V3 a;
    double e;
};
template <int PART>
struct TriXch {
    TriXP X;
    int lane;
    int np, nc;     // values published / consumed (wave-uniform)
    bool dead;      // a poll timed out: no more waiting in this launch
    unsigned dbg_miss = 0, dbg_spin = 0;          // probe builds only (bsk_probes.hpp: TRI_XCHG): finishes whose first read
    unsigned long long dbg_cyc = 0;               // was early; re-reads; cycles spent re-reading
    template <bool EXTRA>
    __device__ __forceinline__ void publish(V3 a, double extra) {
        if constexpr (probe::TRI_NOPUBLISH && PART == PART_TRA) { ++np; return; }   // fault injection (probe builds): the partner times out
        const int slot = np & 3;
        if constexpr (PART == PART_TRA) {
            tri_st(&X->v[slot][0][lane], a.x); tri_st(&X->v[slot][1][lane], a.y); tri_st(&X->v[slot][2][lane], a.z);
            if constexpr (EXTRA) tri_st(&X->v[slot][3][lane], extra);
            tri_order();                               // rows, THEN the tag
            tri_sti(&X->tv[slot][lane], np + 1);
        } else {
            tri_st(&X->s[slot][0][lane], a.x); tri_st(&X->s[slot][1][lane], a.y); tri_st(&X->s[slot][2][lane], a.z);
            tri_order();
            tri_sti(&X->ts[slot][lane], np + 1);
        }
        tri_order();
        ++np;
    }
    template <bool EXTRA>
    __device__ __forceinline__ TriPend prefetch() const {
        const int slot = nc & 3;
        TriPend f;
        f.e = 0.0;
        if constexpr (PART == PART_TRA) {
            f.tag = tri_ldi(&X->ts[slot][lane]);
            tri_order();                               // the tag, THEN the rows
            f.a = mk(tri_ld(&X->s[slot][0][lane]), tri_ld(&X->s[slot][1][lane]), tri_ld(&X->s[slot][2][lane]));
        } else {
            f.tag = tri_ldi(&X->tv[slot][lane]);
            tri_order();
            f.a = mk(tri_ld(&X->v[slot][0][lane]), tri_ld(&X->v[slot][1][lane]), tri_ld(&X->v[slot][2][lane]));
            if constexpr (EXTRA) f.e = tri_ld(&X->v[slot][3][lane]);
        }
        return f;
    }
    template <bool EXTRA>
    __device__ __forceinline__ V3 finish(TriPend f, double& extra) {
        const int want = nc + 1;
        if (BSK_UNLIKELY(__builtin_amdgcn_ballot_w64(f.tag != want) != 0)) {
            if constexpr (probe::XCH_STATS) ++dbg_miss;
            const probe::Stamp c0 = probe::stamp<probe::XCH_STATS>();
            for (int spin = 0; !dead; ++spin) {
                f = prefetch<EXTRA>();
                if constexpr (probe::XCH_STATS) ++dbg_spin;
                if (__builtin_amdgcn_ballot_w64(f.tag != want) == 0) break;
                if (spin > TRI_SPIN_LIMIT) { dead = true; X->err = 1; }
            }
            probe::since<probe::XCH_STATS>(dbg_cyc, c0);
        }
        ++nc;
        if constexpr (EXTRA) extra = f.e;
        return f.a;
    }
};

// ---------------------------------------------------------------------------------------------------------
// Wave-uniform constants as DPP broadcast operands (full-scenario kernels).  These kernels need ~45 doubles that
// are the same in every lane (wheel geometry, facet tables, panel normal, atmosphere) on top of the RK4 loop's
// HotCfg, which alone nearly fills the SGPR file.  Parked in VGPRs they end up in AGPRs (two v_accvgpr_read per
// use); spilled from SGPRs they cost two v_readlane per use; read from LDS their latency is exposed with one wave
// per SIMD.  Every one of them is a multiplicand of an FMA, so they sit in THREE VGPR pairs instead — lane l of
// each 16-lane row holds entry l & 15 — and v_fmac_f64 picks its factor with the DPP control row_newbcast:k: no
// extra instruction, no latency, 6 registers.  A DPP operand is read from ANOTHER lane's register, so every lane
// of the wave must be active where these are used: the full-scenario kernels keep the RK4 loop's trip count and
// the drag / thruster switches wave-uniform.  asm volatile keeps the compiler from sinking one into a branch.
// HAZARD: a DPP read wants two wait states after a VALU write of its source and the compiler does not pad inline asm - the
// build does (csrc/dpp_nops.py between the device compiler and the assembler), tools/dpp_hazard.py checks what ships.
struct KTab {
    double a, b, c, e;
};
template <int K>
__device__ __forceinline__ double fmac_k(double acc, double tab, double x) {        // acc + tab[K] * x
    static_assert(K >= 0 && K < 16, "row lane");
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(K));
    return acc;
}
template <int K>
__device__ __forceinline__ double fmac_k_abs(double acc, double tab, double x) {    // acc + tab[K] * |x|
    asm volatile("v_fmac_f64_dpp %0, %1, |%2| row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(K));
    return acc;
}
template <int K>
__device__ __forceinline__ double fmac_k_neg(double acc, double tab, double x) {    // acc - tab[K] * x
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(K));
    return acc;
}
// max(x, 0) of a value that came out of one of the asm statements above: the compiler cannot see that it is no signalling
// NaN and canonicalises it first (v_max x, x) when asked for fmax
__device__ __forceinline__ double max0(double x) {
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
template <int K>
__device__ __forceinline__ double mul_k(double tab, double x) { return fmac_k<K>(0.0, tab, x); }   // tab[K] * x
template <int K>
__device__ __forceinline__ double get_k(double tab) {                                                // tab[K]
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(tab), "n"(K));
    return r;
}

// exponentialAtmosphere (leoPowerAttitudeSimulator.py:265-271): rho0 exp(-(|r| - Re) / H) from rm = |r|, 0 below the skip
// density (|a_drag| < 1e-19 m/s^2 there).  exp(x) = 2^n e^r with x = n ln2 + r, |r| <= ln2 / 2, and the degree-13 Taylor
// polynomial (truncation 4e-18), its coefficients - scaled by rho0 - in row E of the broadcast table.  A polynomial wants
// its constants as ADDENDS and gfx950 has neither 64-bit literals nor a free scalar register in this loop: left to the
// compiler the thirteen literals are hoisted out of the tick loop and, the register file being full, parked in AGPRs (two
// v_accvgpr_read per use plus the moves that re-pair them: 27 issue slots per tick); materialised in VCC at each use they
// cost two scalar moves each - and with one wave per SIMD a scalar instruction takes an issue slot like any other (measured:
// 81 VALU instructions less, 23 scalar ones more, 3.7 % instead of 9 % faster).  So: seven first-degree pairs c_2j + c_2j+1 r,
// each a broadcast move + a broadcast FMA, combined by Horner's rule in r^2 - 28 instructions, no scalar ones, no constants
// in registers.  Moves first, FMAs after: no DPP instruction directly behind the one that wrote its accumulator.
__device__ __forceinline__ double atmosphere_exponent(const KTab& kt, double rm) {      // Re/H - |r|/H
    return fmac_k<KC_NIH>(get_k<KC_REQIH>(kt.c), kt.c, rm);
}
__device__ __forceinline__ double atmosphere_exp(const KTab& kt, double x) {             // rho0 e^x
    const double n = __builtin_rint(mul_k<KC_LOG2E>(kt.c, x));
    const double te = kt.e;
    double r = fmac_k<KE_NLN2HI>(x, te, n);
    double t0 = get_k<KE_POLY + 0>(te), t1 = get_k<KE_POLY + 2>(te), t2 = get_k<KE_POLY + 4>(te), t3 = get_k<KE_POLY + 6>(te);
    r = fmac_k<KE_NLN2LO>(r, te, n);
    double t4 = get_k<KE_POLY + 8>(te), t5 = get_k<KE_POLY + 10>(te), t6 = get_k<KE_POLY + 12>(te);
    const double r2 = r * r;
    t0 = fmac_k<KE_POLY + 1>(t0, te, r); t1 = fmac_k<KE_POLY + 3>(t1, te, r); t2 = fmac_k<KE_POLY + 5>(t2, te, r);
    t3 = fmac_k<KE_POLY + 7>(t3, te, r); t4 = fmac_k<KE_POLY + 9>(t4, te, r); t5 = fmac_k<KE_POLY + 11>(t5, te, r);
    t6 = fmac_k<KE_POLY + 13>(t6, te, r);
    double p = fma(t6, r2, t5);
    p = fma(p, r2, t4);
    p = fma(p, r2, t3);
    p = fma(p, r2, t2);
    p = fma(p, r2, t1);
    p = fma(p, r2, t0);
    return ldexp(p, (int)n);
}
// The density ALONG a trajectory.  Between two dyn ticks |r| moves by metres and the exponent by d = -(|r'| - |r|) / H: up to
// 5e-3 on the reference's orbits (radial velocity up to 380 m/s, H = 8 km, dt = 0.1 s).  rho' = rho e^d with the degree-6
// Taylor polynomial of e^d (truncation d^7 / 5040 < 5e-17 for |d| <= 2^-6, i.e. 125 m of radial motion per tick) - ten
// instructions instead of the thirty-five of the full evaluation.  The exponents telescope exactly (x' - x of neighbouring
// doubles is exact), so what accumulates is the rounding of one polynomial and one product per tick (2e-16 each); every
// chunk of <= PEN_CHUNK ticks starts from a full evaluation (anchor), and a lane whose exponent jumps by more than 2^-6
// takes the full evaluation for that tick (per lane: the others keep their increments, so a lane's result does not depend
// on its neighbours in the wave).  The conditioning of the exponential itself - an ulp of |r| is 1e-13 of rho - is three
// orders above either.  All three forms of the scenario kernel anchor at the same ticks and advance with the same
// operations: bit-identical.
struct Atmo {
    double x, rho;             // the exponent of the last evaluation, the un-clamped density there
    __device__ __forceinline__ double clamped(const KTab& kt) const { return rho >= get_k<KC_RSKIP>(kt.c) ? rho : 0.0; }   // 0 below the skip density
    __device__ __forceinline__ double anchor(const KTab& kt, double rm) {
        x = atmosphere_exponent(kt, rm);
        rho = atmosphere_exp(kt, x);
        return clamped(kt);
    }
    __device__ __forceinline__ double advance(const KTab& kt, double rm) {
        const double xn = atmosphere_exponent(kt, rm);
        const double d = xn - x;
        x = xn;
        const double d2 = d * d;
        double a = fmac_k<KC_I120>(get_k<KC_I24>(kt.c), kt.c, d);          // 1/24 + d/120 + d^2/720
        double b = fmac_k<KC_I6>(0.5, kt.c, d);                             // 1/2 + d/6
        a = fmac_k<KC_I720>(a, kt.c, d2);
        b = fma(d2, a, b);
        const double pn = rho * fma(d2, b, d + 1.0);
        const bool far = !(d2 <= 0x1p-12);                                   // |d| > 2^-6 or NaN
        if (BSK_UNLIKELY(__builtin_amdgcn_ballot_w64(far) != 0)) {
            const double full = atmosphere_exp(kt, xn);
            rho = far ? full : pn;
        } else rho = pn;
        return clamped(kt);
    }
};
// exponentialAtmosphere at one position (no history)
__device__ __forceinline__ double atmosphere_density(const KTab& kt, double rm) {
    const double rho = atmosphere_exp(kt, atmosphere_exponent(kt, rm));
    return rho >= get_k<KC_RSKIP>(kt.c) ? rho : 0.0;
}

// wheel geometry through the broadcast table (rows A and B of KTab): the same operations in the same order per
// accumulator as WheelV, issued wheel-interleaved so that no DPP FMA follows the instruction that produced one of
// its operands (the compiler pads that distance with s_nop, and every s_nop costs a one-wave SIMD an issue slot)
// FOLD (the full-scenario levels): the wheel inertia folded into the table.  The LDS-scratch level keeps the unfolded
// form, whose operations per accumulator are those of WheelV - it is held bit-identical to the register kernel.
template <int NRW, bool FOLD>
struct WheelDpp;
template <int NRW>
struct WheelDppBase {
    double ta, tb, td;
    __device__ __forceinline__ void tail(V3 dw, const double* base, double* Om) const {
#pragma unroll
        for (int i = 0; i < NRW; ++i) Om[i] = base[i];
        if constexpr (NRW > 0) Om[0] = fmac_k_neg<KA_G + 2>(Om[0], ta, dw.z);
        if constexpr (NRW > 1) Om[1] = fmac_k_neg<KA_G + 5>(Om[1], ta, dw.z);
        if constexpr (NRW > 2) Om[2] = fmac_k_neg<KA_G + 8>(Om[2], ta, dw.z);
        if constexpr (NRW > 3) Om[3] = fmac_k_neg<KA_G + 11>(Om[3], ta, dw.z);
        if constexpr (NRW > 0) Om[0] = fmac_k_neg<KA_G + 1>(Om[0], ta, dw.y);
        if constexpr (NRW > 1) Om[1] = fmac_k_neg<KA_G + 4>(Om[1], ta, dw.y);
        if constexpr (NRW > 2) Om[2] = fmac_k_neg<KA_G + 7>(Om[2], ta, dw.y);
        if constexpr (NRW > 3) Om[3] = fmac_k_neg<KA_G + 10>(Om[3], ta, dw.y);
        if constexpr (NRW > 0) Om[0] = fmac_k_neg<KA_G + 0>(Om[0], ta, dw.x);
        if constexpr (NRW > 1) Om[1] = fmac_k_neg<KA_G + 3>(Om[1], ta, dw.x);
        if constexpr (NRW > 2) Om[2] = fmac_k_neg<KA_G + 6>(Om[2], ta, dw.x);
        if constexpr (NRW > 3) Om[3] = fmac_k_neg<KA_G + 9>(Om[3], ta, dw.x);
    }
};
template <int NRW>
struct WheelDpp<NRW, true> : WheelDppBase<NRW> {
    using WheelDppBase<NRW>::ta;
    using WheelDppBase<NRW>::tb;
    using WheelDppBase<NRW>::td;
    // The wheel inertia folded into the table (row D: Js_i g_i and dt / Js_i): the momentum sum reads the wheel speeds
    // directly and the end-of-step base is one DPP FMA per wheel - no Js Om_i / tq_i / Js_i intermediates, each of
    // which cost a zero-initialising move and a one-term chain (16 issue slots per step less)
    __device__ __forceinline__ void head(const double* tq, const double* Om, V3& T, V3& p, double* tqj) const {
        T = mk(0, 0, 0);        // (p comes in as W w0 and takes the wheels' momentum on top)
        auto wheel = [&](auto IC) {
            constexpr int i = decltype(IC)::value;
            T.x = fmac_k<KA_G + 3 * i>(T.x, ta, tq[i]); T.y = fmac_k<KA_G + 3 * i + 1>(T.y, ta, tq[i]); T.z = fmac_k<KA_G + 3 * i + 2>(T.z, ta, tq[i]);
            p.x = fmac_k<KD_JG + 3 * i>(p.x, td, Om[i]); p.y = fmac_k<KD_JG + 3 * i + 1>(p.y, td, Om[i]); p.z = fmac_k<KD_JG + 3 * i + 2>(p.z, td, Om[i]);
        };
        if constexpr (NRW > 0) wheel(std::integral_constant<int, 0>{});
        if constexpr (NRW > 1) wheel(std::integral_constant<int, 1>{});
        if constexpr (NRW > 2) wheel(std::integral_constant<int, 2>{});
        if constexpr (NRW > 3) wheel(std::integral_constant<int, 3>{});
    }
    __device__ __forceinline__ void bases(double h, const double* tq, const double* tqj, const double* Om, double* base) const {
#pragma unroll
        for (int i = 0; i < NRW; ++i) base[i] = Om[i];
        if constexpr (NRW > 0) base[0] = fmac_k<KD_HIJS + 0>(base[0], td, tq[0]);
        if constexpr (NRW > 1) base[1] = fmac_k<KD_HIJS + 1>(base[1], td, tq[1]);
        if constexpr (NRW > 2) base[2] = fmac_k<KD_HIJS + 2>(base[2], td, tq[2]);
        if constexpr (NRW > 3) base[3] = fmac_k<KD_HIJS + 3>(base[3], td, tq[3]);
    }
};
template <int NRW>
struct WheelDpp<NRW, false> : WheelDppBase<NRW> {
    using WheelDppBase<NRW>::ta;
    using WheelDppBase<NRW>::tb;
    using WheelDppBase<NRW>::td;
    __device__ __forceinline__ void bases(double h, const double* tq, const double* tqj, const double* Om, double* base) const {
#pragma unroll
        for (int i = 0; i < NRW; ++i) base[i] = fma(h, tqj[i], Om[i]);
    }
    __device__ __forceinline__ void head(const double* tq, const double* Om, V3& T, V3& p, double* tqj) const {
        double jo[NRW > 0 ? NRW : 1];
        T = mk(0, 0, 0);        // (p comes in as W w0)
#pragma unroll
        for (int i = 0; i < NRW; ++i) { jo[i] = 0.0; tqj[i] = 0.0; }
        if constexpr (NRW > 0) jo[0] = fmac_k<KA_JS + 0>(jo[0], ta, Om[0]);
        if constexpr (NRW > 1) jo[1] = fmac_k<KA_JS + 1>(jo[1], ta, Om[1]);
        if constexpr (NRW > 2) jo[2] = fmac_k<KA_JS + 2>(jo[2], ta, Om[2]);
        if constexpr (NRW > 3) jo[3] = fmac_k<KA_JS + 3>(jo[3], ta, Om[3]);
        if constexpr (NRW > 0) tqj[0] = fmac_k<KB_IJS + 0>(tqj[0], tb, tq[0]);
        if constexpr (NRW > 1) tqj[1] = fmac_k<KB_IJS + 1>(tqj[1], tb, tq[1]);
        if constexpr (NRW > 2) tqj[2] = fmac_k<KB_IJS + 2>(tqj[2], tb, tq[2]);
        if constexpr (NRW > 3) tqj[3] = fmac_k<KB_IJS + 3>(tqj[3], tb, tq[3]);
        auto wheel = [&](auto IC) {     // six independent accumulators per wheel
            constexpr int i = decltype(IC)::value;
            T.x = fmac_k<KA_G + 3 * i>(T.x, ta, tq[i]); T.y = fmac_k<KA_G + 3 * i + 1>(T.y, ta, tq[i]); T.z = fmac_k<KA_G + 3 * i + 2>(T.z, ta, tq[i]);
            p.x = fmac_k<KA_G + 3 * i>(p.x, ta, jo[i]); p.y = fmac_k<KA_G + 3 * i + 1>(p.y, ta, jo[i]); p.z = fmac_k<KA_G + 3 * i + 2>(p.z, ta, jo[i]);
        };
        if constexpr (NRW > 0) wheel(std::integral_constant<int, 0>{});
        if constexpr (NRW > 1) wheel(std::integral_constant<int, 1>{});
        if constexpr (NRW > 2) wheel(std::integral_constant<int, 2>{});
        if constexpr (NRW > 3) wheel(std::integral_constant<int, 3>{});
    }
};

// The two coefficients of the MRP rotation  [BN] = I + (8 s~^2 - 4 (1 - s^2) s~) / (1 + s^2)^2:  ka = 8 / (1 + s^2)^2,
// kb = 4 (1 - s^2) / (1 + s^2)^2 (12 instructions, one of them a quarter-rate reciprocal).  One definition for every place
// that rotates with an attitude (panel, drag, thrusters), so that a value computed at one site can stand in at another.
struct MrpRot {
    double ka, kb;
};
__device__ __forceinline__ MrpRot mrp_rot_q2(double q2) {                    // q2 = |sigma|^2
    const double op = 1.0 + q2, iop2 = rcp_nr(op * op);
    return MrpRot{8.0 * iop2, 4.0 * (1.0 - q2) * iop2};
}
__device__ __forceinline__ MrpRot mrp_rot(V3 sig) { return mrp_rot_q2(dot(sig, sig)); }
// What the end of one dyn tick already knows about the state the next tick starts from (full-scenario single-wave
// kernels): the EnvTask evaluates |r|^2 (eclipse) and the rotation coefficients of the attitude (panel) of exactly the
// state whose first integrator stage needs both again (gravity; drag / thrust rotation), and the atmosphere needs 1/|r|
// (ir: filled in by the tick's head).  Carried across the tick boundary instead of being computed twice: 21 instructions
// and two quarter-rate ones per tick less; the same operations on the same operands, so the same bits.
struct Pre {
    double r2, ir;
    MrpRot rot;
};

// shadow factor where it is cheap (1 lit, 0 umbra), `band` where the disc is partially covered; r2 = |r|^2
__device__ __forceinline__ double shadow_quick(const SunGeom& g, V3 r, double r2, bool& band) {
    // branch-free: with 64 spacecraft per wave every path is taken by some lane anyway, and each divergent branch
    // costs exec-mask bookkeeping on the scalar unit
    const double rs = dot(r, g.sun);
    const double s0 = -rs * g.ism;
    const double c1 = s0 + g.re_sf1, c2 = s0 - g.re_sf2;
    const double l2v = fma(-s0, s0, r2);                 // squared distance from the shadow axis
    const double l1 = c1 * g.tf1, l2 = c2 * g.tf2;
    const bool night = !(r2 < 2.0 * rs);                 // not on the day side of the planet
    const bool in2 = l2v < l2 * l2, in1 = l2v < l1 * l1;
    const bool umbra = night && in2 && c2 < 0.0;         // inside the umbra cone: the disc is fully covered
    band = night && !umbra && (in2 || in1);              // penumbra / antumbra band
    return umbra ? 0.0 : 1.0;
}

// eclipse (classification) -> simpleSolarPanel of one dyn tick: the panel power per unit of lit disc (`gain`), the shadow
// factor where it is cheap, `band` where the disc is partially covered.  One definition for the single-wave kernels' tick
// record and the pair form's environment wave: the same operations in the same order.
template <bool LDSK>
__device__ __forceinline__ void power_eval(const PowerCfg& pc, const SunGeom& g, V3 r, V3 sig, double tc, double& gain, double& sh, bool& band,
                                           Pre* next = nullptr, const double* q2_known = nullptr) {
    const double r2 = dot(r, r);
    sh = shadow_quick(g, r, r2, band);
    const V3 d = g.sun - r;
    const double d2 = dot(d, d), id = rsqrt_nr(d2);
    // dB = [BN] d with the Sun vector left at its length: the panel's cosine is n . dB / |d|, and the flux falls with
    // 1 / |d|^2, so the three products that would normalise d first fold into one power of 1 / |d|
    const MrpRot rot = mrp_rot_q2(q2_known ? *q2_known : dot(sig, sig));   // (|sigma|^2 of this attitude: the RK4 step's shadow-set test has it)
    if (next) { next->r2 = r2; next->rot = rot; }
    const V3 t1 = cross(sig, d), t2 = cross(sig, t1);
    const V3 dB = d + rot.ka * t2 - rot.kb * t1;
    const double id3 = id * id * id;
    if constexpr (LDSK) {     // panel normal and flux constant from the broadcast table (row C)
        const double proj = max0(fmac_k<KC_NB>(fmac_k<KC_NB + 1>(mul_k<KC_NB + 2>(tc, dB.z), tc, dB.y), tc, dB.x));
        gain = mul_k<KC_KFLUX>(tc, id3) * proj;
    } else {
        const double proj = fmax(fma(pc.nB[0], dB.x, fma(pc.nB[1], dB.y, pc.nB[2] * dB.z)), 0.0);
        gain = pc.kflux * id3 * proj;
    }
}

// per tick: classify, record the panel gain and (when known) the shadow factor of slot t
template <bool LDSK>
__device__ __forceinline__ void power_tick(const PowerCfg& pc, const SunGeom& g, V3 r, V3 sig, LdsP L, int t, int lane, double tc, Pre* next = nullptr,
                                           const double* q2_known = nullptr) {
    bool band;
    double sh, gain;
    power_eval<LDSK>(pc, g, r, sig, tc, gain, sh, band, next, q2_known);
    L->g[t][lane] = gain;
    if (BSK_UNLIKELY(band)) {
        const int e = __hip_atomic_fetch_add(&L->qcount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (BSK_LIKELY(e < PEN_QCAP)) {
            L->qr[0][e] = r.x; L->qr[1][e] = r.y; L->qr[2][e] = r.z;
            L->qown[e] = lane | (t << 8);
        } else {
            L->s[t][lane] = percent_shadow(pc, g.sun - r, r, dot(r, r));   // queue full: this lane evaluates its own tick now
        }
    } else {
        L->s[t][lane] = sh;
    }
}

// after the chunk's ticks (control flow reconverged, every lane of the wave here): drain the queue cooperatively,
// then replay this lane's `m` battery updates in order.  `shadow` ends as the last tick's factor.
__device__ __forceinline__ void power_flush(const PowerCfg& pc, LdsP L, int m, int lane, double h, double& charge,
                                            double& shadow) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // The queue counter and the whole record are read in ONE batch (2 x PEN_SLOTS + 1 LDS reads in flight): the wave
    // is alone on its SIMD and waits every round trip out.  Only when the queue holds something (rare) the record's
    // shadow factors are read again after the drain has filled them in.
    const double draw = pc.draw, cap = pc.cap;
    double sk[PEN_SLOTS], dq[PEN_SLOTS];
    const int qc = min(L->qcount, PEN_QCAP);               // (pushes beyond the capacity were evaluated in place)
#pragma unroll
    for (int k = 0; k < PEN_SLOTS; ++k) {
        sk[k] = L->s[k][lane];
        dq[k] = L->g[k][lane];
    }
    if (BSK_UNLIKELY(qc > 0)) {                           // wave-uniform
        for (int e = lane; e < qc; e += 64) {
            const int own = L->qown[e], ol = own & 63, k = own >> 8;
            const V3 r = mk(L->qr[0][e], L->qr[1][e], L->qr[2][e]);
            const V3 sun = mk(L->sun[0][ol], L->sun[1][ol], L->sun[2][ol]);
            L->s[k][ol] = percent_shadow(pc, sun - r, r, dot(r, r));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) L->qcount = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < PEN_SLOTS; ++k) sk[k] = L->s[k][lane];
    }
    // Replay: the panel powers p_k in parallel, then only the clamped sums are a dependent chain (three operations per tick;
    // one wave per SIMD pays every dependent instruction's full latency).  charge' = clamp(fma(p, h, charge)): ONE rounding
    // per tick, written as an explicit fma - under -ffp-contract=fast the compiler fused `charge + p * h` in some shapes of
    // this loop and not in others, and the wave-split forms' replay (bsk_kernels.hip: env_ticks) must give the same bits.
    // (The reference's simpleBattery rounds p h first: <= 1 ulp of the charge per tick apart, 1e-16 relative.)
#pragma unroll
    for (int k = 0; k < PEN_SLOTS; ++k) dq[k] = fma(dq[k], sk[k], draw);
    // `m` is the same in every lane (the chunk lengths it sums are wave-uniform, the flush is decided by a ballot): as a
    // scalar, the full record - the usual flush - replays without a predicate per slot (three operations instead of eight)
    const int mu = __builtin_amdgcn_readfirstlane(m);
    if (BSK_LIKELY(mu == PEN_SLOTS)) {
#pragma unroll
        for (int k = 0; k < PEN_SLOTS; ++k) charge = fmin(fmax(fma(dq[k], h, charge), 0.0), cap);
        shadow = sk[PEN_SLOTS - 1];
    } else {
#pragma unroll
        for (int k = 0; k < PEN_SLOTS; ++k) {
            if (k < mu) {
                shadow = sk[k];
                charge = fmin(fmax(fma(dq[k], h, charge), 0.0), cap);
            }
        }
    }
}

template <bool DIAG>
__device__ __forceinline__ V3 mv3(const double* m, V3 v) {
    if constexpr (DIAG) return V3{m[0] * v.x, m[1] * v.y, m[2] * v.z};
    else return mv(m, v);
}

template <int NRW>
struct State {
    V3 r, v, s, w;  // position, velocity, sigma_BN, omega_BN_B
    double Om[NRW > 0 ? NRW : 1];
};

// --------------------------------------------------------------------------------------------
// Pines' normalised spherical-harmonic field (SURVEY.md §8 note N1), degree d, position in the
// planet-fixed frame.  All lanes walk the same (l, m) sequence, so the coefficient stream is
// wave-uniform and is read with SCALAR loads: the host fuses Cbar/Sbar/n1/n2/nq1/nq2 into one
// stream in iteration order (bsk_capi.hip: build_sh_table), 8 doubles = one s_load_dwordx16 per
// step, and the VALU takes them as SGPR operands.  Columns M = 1..d+1 of the derived Legendre
// function A[L][M], L = M..d+1, are generated by the three-term recursion on B = w_L A[L][M]
// (w_L = mu/(r Re) (Re/r)^(L+1) folded into the recursion), and each column's six coefficient
// sums are combined with (Re, Im)(s + i t)^(M-1) once per column: 10 fp64 ops per (L, M).
typedef const double __attribute__((address_space(4))) * CTab;

// SPLIT (form of the harmonics evaluation): 1 = stream read with scalar loads (the first version, kept
// for comparison), 4 = stream read with coalesced VECTOR loads and fed to the VALU through the DPP row
// broadcast (gravity_sh_dpp), 5 = the same with each spacecraft's columns split over two cooperating
// waves (two waves per SIMD at 65 536 spacecraft).  Two further forms were measured and dropped
// (scalar stream split over two waves, whole stream resident in LDS; code at commit 28b481e).

// acc += (lane L of each 16-lane row of `tab`) * b: v_fmac_f64 taking its first factor through the
// DPP row_newbcast control, i.e. a wave-uniform table value costs the VALU nothing beyond the FMA
// itself and needs neither SGPRs nor the scalar cache.  Every lane of the wave must be active (the
// broadcast reads a register of another lane): callers keep control flow wave-uniform.
// asm volatile pins the instruction in program order: the harmonics walk fixes its own order (dependent
// fp64 ops kept apart by independent ones), which the scheduler would undo.
template <int L>
__device__ __forceinline__ double fmac_bc(double acc, double tab, double b) {
    static_assert(L >= 0 && L < 16, "row lane");
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(b), "n"(L));
    return acc;
}

// chunks in flight = chunks per loop body (the host pads the stream to whole bodies + slack)
constexpr int SH_RING = 8;

// Forms 4 / 5 of the harmonics evaluation.  The fused stream (bsk_capi.hip: build_sh_table_dpp) holds 8
// doubles per (L, M) entry in iteration order, every column padded to an even number of entries, so a
// 128-byte chunk = 2 entries = the 16 doubles one 16-lane row holds in ONE VGPR pair: lane l of every
// row loads double (l & 15) of the chunk (a single global_load_dwordx2, one cache line per wave), and
// the FMAs pick their table operand with row_newbcast.  Chunks are prefetched SH_RING ahead in a ring of
// SH_RING registers (the loop body is SH_RING chunks, so the ring needs no register moves); the stream is
// a plain linear walk, the only branch per chunk is the (rare, wave-uniform) column end.
// The three-term recursion runs on Bt = B / alpha_L with alpha_L = n2_L alpha_(L-2) folded into the
// table (entry[0] = n1_L alpha_(L-1) / alpha_L, coefficient products carry alpha_L), which leaves
//   Bt_L = entry[0] (u rho) Bt_(L-1) - rho^2 Bt_(L-2)                     3 fp64 ops
// plus the six coefficient sums = 9 fp64 ops per (L, M) entry, no SGPR operands.
// With one wave per SIMD every instruction of any kind costs the wave a full issue slot (measured: 18
// VALU + 7 other instructions per chunk = 60 % VALU-active), so at small batches the columns of each
// spacecraft are split over two waves (TWO; a 256-thread workgroup = 2 x 64 spacecraft x 2 halves, so that
// its four waves land on the four SIMDs of a CU): both waves of a pair carry the same 64 spacecraft,
// each walks half of the entries, they exchange four partial sums through LDS, and the SIMD overlaps
// one wave's loads / scalar instructions with the other's FMAs.  Both forms add the two halves' partial
// sums in the same order, so their results are bit-identical.
template <bool TWO, class Hot>
__device__ __forceinline__ V3 gravity_sh_dpp(const Hot& c, V3 p) {
    const double r2 = dot(p, p);
    const double ir = rsqrt_nr(r2);
    const double s = p.x * ir, t = p.y * ir, u = p.z * ir;
    const double rho = c.req * ir;              // Re / r
    const double irho = r2 * ir * c.inv_req;    // r / Re
    const double w0 = c.mu_over_req * ir * rho; // mu/(r Re) * (Re/r)
    const double ur = u * rho, nrr = -(rho * rho);
    const int d1 = c.sh_degree + 1;
    const int split = c.sh_split;
    int half = 0;
    if constexpr (TWO) half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 1);   // wave-uniform
    const int m_lo = (TWO && half) ? split : 1;
    const int m_hi = (TWO && !half) ? split : d1 + 1;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = (half == 0) ? -w0 : 0.0;
    double b1 = 0.0, b2 = 0.0, b3 = 0.0, b4 = 0.0;   // first half's partial sums (one-wave form)
    // (Re, Im)(s + i t)^(m_lo - 1) and w_(m_lo)
    double cr = 1.0, ci = 0.0, wM = w0 * rho;
    for (int M = 1; M < m_lo; ++M) {
        const double ncr = fma(s, cr, -t * ci);
        ci = fma(s, ci, t * cr);
        cr = ncr;
        wM *= rho;
    }
    // recursion state: P = Bt_(L-1), Bn = -rho^2 Bt_(L-2), m1 = (u rho) Bt_(L-1).  A column starts from
    // P = Bn = 0 and m1 = w_M, which makes the generic step produce Bt_M = entry[0] w_M (the diagonal
    // constant) and Bt_(M+1) = entry[0] (u rho) Bt_M with no special case.
    double P = 0.0, m1 = wM;
    double X1 = 0.0, X2 = 0.0, Y1 = 0.0, Y2 = 0.0, Z1 = 0.0, Z2 = 0.0;

    const uint32_t lo = (threadIdx.x & 15u) * 8u;
    const double* tb = c.sh_tab + ((TWO && half) ? (int64_t)16 * c.sh_chunk1 : 0);
    // the ring loads must be issued in the loop's refill order and before the loop (the scheduler would
    // otherwise reverse them and sink the last one into the loop, and the vmcnt state merged from the loop
    // entry and the back edge then over-waits on every iteration): a compiler barrier after each
    double q[SH_RING];
    q[SH_RING - 1] = ldf(tb, lo);    // only read with a zero factor before its first refill; issued first
    asm volatile("" ::: "memory");   // so that the ring order at loop entry equals the order in the loop
#pragma unroll
    for (int k = 0; k < SH_RING - 1; ++k) {
        q[k] = ldf(tb + 16 * k, lo);
        asm volatile("" ::: "memory");
    }
    tb += 16 * SH_RING;
    int M = m_lo, rem = (d1 - m_lo + 2) >> 1;   // chunks of column M: (d1 - M + 2) / 2

    // One entry = the recursion step (two dependent ops: fmac -> mul -> next entry's fmac) interleaved with
    // the six coefficient sums of the PREVIOUS entry, so that no fp64 op waits on its predecessor's result.
    // The DPP FMAs are pinned in this order (asm volatile); the two plain multiplies are left to the compiler's
    // scheduler, which places them so that no hazard padding is needed (written as pinned asm, in the slots the
    // distances seem to ask for, each chunk carried two s_nop: 267 -> 256 us at 65 536 spacecraft).
    // Bn = -rho^2 Bt_(L-2) is prepared one entry ahead; the entry opens with the instruction that needs
    // the chunk just loaded, so one s_waitcnt serves the whole chunk.
    // Bp = the previous entry's Bt (its sums are still pending), (qp, OP) = where its coefficients sit.
    double Bp = 0.0, Bn = 0.0;
    auto entry = [&](double qc, auto OC, double qp, auto OP) {
        constexpr int oc = decltype(OC)::value, op = decltype(OP)::value;
        const double Pold = P;
        const double B = fmac_bc<oc + 0>(Bn, qc, m1);
        X1 = fmac_bc<op + 2>(X1, qp, Bp); X2 = fmac_bc<op + 3>(X2, qp, Bp);
        Bn = nrr * Pold;
        Y1 = fmac_bc<op + 4>(Y1, qp, Bp); Y2 = fmac_bc<op + 5>(Y2, qp, Bp);
        m1 = ur * B;
        Z1 = fmac_bc<op + 6>(Z1, qp, Bp); Z2 = fmac_bc<op + 7>(Z2, qp, Bp);   // two before the next entry reads m1
        P = B;
        Bp = B;
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    auto column_end = [&](double qc) {
        // flush the last entry's sums, combine with (Re, Im)(s + i t)^(M-1), advance to (s + i t)^M, restart
        X1 = fmac_bc<10>(X1, qc, Bp); X2 = fmac_bc<11>(X2, qc, Bp); Y1 = fmac_bc<12>(Y1, qc, Bp);
        Y2 = fmac_bc<13>(Y2, qc, Bp); Z1 = fmac_bc<14>(Z1, qc, Bp); Z2 = fmac_bc<15>(Z2, qc, Bp);
        a1 = fma(cr, X1, fma(ci, X2, a1));
        a2 = fma(cr, X2, fma(-ci, X1, a2));
        a3 = fma(cr, Y1, fma(ci, Y2, a3));
        a4 = fma(-irho, fma(cr, Z1, ci * Z2), a4);
        const double ncr = fma(s, cr, -t * ci);
        ci = fma(s, ci, t * cr);
        cr = ncr;
        wM *= rho;
        m1 = wM;
        P = 0.0; Bn = 0.0; Bp = 0.0;
        X1 = 0.0; X2 = 0.0; Y1 = 0.0; Y2 = 0.0; Z1 = 0.0; Z2 = 0.0;
        ++M;
        if constexpr (!TWO) {
            if (M == split) {   // the second half's sums start from zero, as in the two-wave form
                b1 = a1; b2 = a2; b3 = a3; b4 = a4;
                a1 = 0.0; a2 = 0.0; a3 = 0.0; a4 = 0.0;
            }
        }
        // 0 after the last column of this walk: no further column end fires (the chunks that pad the walk
        // to a whole body then only feed sums that are never combined)
        rem = (M < m_hi) ? (d1 - M + 2) >> 1 : 0;
    };
    // tb points at the body after the current one
    const int bodies = TWO ? (half ? c.sh_bodies1 : c.sh_bodies0) : c.sh_bodies;
    for (int body = bodies; body > 0; --body) {
#pragma unroll
        for (int k = 0; k < SH_RING; ++k) {
            constexpr int R = SH_RING;
            const int kp = (k + R - 1) % R;
            entry(q[k], I0{}, q[kp], I8{});
            // the previous chunk's register is free now: refill it (chunk kp of the next body; for k == 0
            // the last chunk of this body)
            q[kp] = ldf(tb + (k == 0 ? -16 : 16 * kp), lo);
            entry(q[k], I8{}, q[k], I0{});
            if (__builtin_expect(--rem == 0, 0)) column_end(q[k]);
        }
        tb += 16 * SH_RING;
    }
    if constexpr (TWO) {
        // exchange the two halves' partial sums through LDS; both waves add them in the same order, so
        // both continue with bit-identical accelerations
        __shared__ double part[2][2][4][64];
        const int lane = threadIdx.x & 63, grp = threadIdx.x >> 7;
        part[grp][half][0][lane] = a1; part[grp][half][1][lane] = a2; part[grp][half][2][lane] = a3; part[grp][half][3][lane] = a4;
        __syncthreads();
        a1 = part[grp][0][0][lane] + part[grp][1][0][lane];
        a2 = part[grp][0][1][lane] + part[grp][1][1][lane];
        a3 = part[grp][0][2][lane] + part[grp][1][2][lane];
        a4 = part[grp][0][3][lane] + part[grp][1][3][lane];
        __syncthreads();
    } else {
        a1 = b1 + a1; a2 = b2 + a2; a3 = b3 + a3; a4 = b4 + a4;
    }
    return V3{fma(s, a4, a1), fma(t, a4, a2), fma(u, a4, a3)};
}

// Form 1: the scalar-load stream (first version; kept selectable for comparison, BSKGPU_SH_FORM=1).
template <class Hot>
__device__ __forceinline__ V3 gravity_sh(const Hot& c, V3 p) {
    const double r2 = dot(p, p);
    const double ir = rsqrt_nr(r2);
    const double s = p.x * ir, t = p.y * ir, u = p.z * ir;
    const double rho = c.req * ir;              // Re / r
    const double irho = r2 * ir * c.inv_req;    // r / Re
    const double w0 = c.mu_over_req * ir * rho; // mu/(r Re) * (Re/r)
    const double ur = u * rho, rr = rho * rho;
    const int d1 = c.sh_degree + 1;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = -w0;
    double cr = 1.0, ci = 0.0, wM = w0;
    double B1 = 0.0, B2 = 0.0, X1 = 0.0, X2 = 0.0, Y1 = 0.0, Y2 = 0.0, Z1 = 0.0, Z2 = 0.0;

    // recursion step A[L][M] <- A[L-1][M], A[L-2][M] and the six coefficient sums (10 fp64 ops)
    auto rec = [&](auto q) {
        const double B = fma(ur, q[0] * B1, -rr * (q[1] * B2));
        B2 = B1;
        B1 = B;
        X1 = fma(B, q[2], X1); X2 = fma(B, q[3], X2);
        Y1 = fma(B, q[4], Y1); Y2 = fma(B, q[5], Y2);
        Z1 = fma(B, q[6], Z1); Z2 = fma(B, q[7], Z2);
    };

    // The stream is read with plain scalar loads, four entries (4 x s_load_dwordx16, 64 SGPRs) per
    // wait, so one scalar-cache round trip is amortised over 40 fp64 ops.  (A two-tuple software
    // pipeline through inline-asm loads was tried: hipcc copies an in-flight tuple at the loop
    // back-edge before its wait, which reads SGPRs the load has not written yet.)
    CTab e = (CTab)c.sh_tab;
    for (int M = 1; M <= d1; ++M) {
        wM *= rho;                               // column start: A[M][M] is the diagonal constant
        B1 = wM * e[0];
        B2 = 0.0;
        X1 = B1 * e[2]; X2 = B1 * e[3]; Y1 = B1 * e[4]; Y2 = B1 * e[5]; Z1 = B1 * e[6]; Z2 = B1 * e[7];
        e += 8;
        int n = d1 - M;                          // entries left in this column
        for (; n >= 4; n -= 4) {
            rec(e); rec(e + 8); rec(e + 16); rec(e + 24);
            e += 32;
        }
        for (; n > 0; --n) {
            rec(e);
            e += 8;
        }
        // column end: combine with (Re, Im)(s + i t)^(M-1), advance to (s + i t)^M
        a1 = fma(cr, X1, fma(ci, X2, a1));
        a2 = fma(cr, X2, fma(-ci, X1, a2));
        a3 = fma(cr, Y1, fma(ci, Y2, a3));
        a4 = fma(-irho, fma(cr, Z1, ci * Z2), a4);
        const double ncr = fma(s, cr, -t * ci);
        ci = fma(s, ci, t * cr);
        cr = ncr;
    }
    return V3{fma(s, a4, a1), fma(t, a4, a2), fma(u, a4, a3)};
}

// --------------------------------------------------------------------------------------------
// gravity: point mass (+ closed-form J2): 15 / 22 fp64 ops; spherical harmonics above.
// tsim is only used by the harmonics (planet rotation about the inertial z axis).
// third-body perturbation relative to the central body: a3(r) = mu_s [ (s - r)/|s - r|^3 - s/|s|^3 ].
// Evaluated exactly once per chunk of <= PEN_SLOTS ticks, at the chunk's first position r_c, and carried through
// the RK4 stages by its gradient, the tidal tensor G = k (3 s^ s^T - 1), k = mu_s/|s|^3 (per launch, about the
// planet's centre):  a3(r) ~ a3(r_c) + G (r - r_c) = A0 + G r.  What is dropped: the second-order term,
// ~3 mu_s |r - r_c|^2 / |s|^4 = 5e-17 m/s^2 for the 7.6 km a chunk covers, and G's change between the planet's
// centre and r_c, 3 |r|/|s| of the first-order term = 4e-14 m/s^2; times h that is 1/200 of one ulp of the velocity
// per step (the exact per-stage evaluation, 22 instructions against 10, is what the oracle does).
struct Sun3 {
    V3 sh;        // unit vector to the Sun (per lane: each spacecraft's own clock)
    double k;     // mu_s / |s|^3
    V3 A0;        // a3(r_c) - G r_c
};
__device__ __forceinline__ V3 third_body_exact(V3 sun, double mu, double k, V3 r) {
    const V3 d = sun - r;
    const double id = rsqrt_nr(dot(d, d));
    const V3 a = (mu * id * id * id) * d;
    return mk(fma(-k, sun.x, a.x), fma(-k, sun.y, a.y), fma(-k, sun.z, a.z));
}
// k (3 (s^.r) s^ - r)
__device__ __forceinline__ V3 tidal(const Sun3& s3, V3 r, V3 base) {
    const double t = 3.0 * dot(s3.sh, r);
    return mk(fma(s3.k, fma(t, s3.sh.x, -r.x), base.x), fma(s3.k, fma(t, s3.sh.y, -r.y), base.y),
              fma(s3.k, fma(t, s3.sh.z, -r.z), base.z));
}
// The third body's acceleration of one RK4 step: A0 + G r_m at the step's midpoint estimate r_m = r + h/2 v, held through
// the four stages.  The tidal term is linear in r, so RK4's weights integrate its variation along the step exactly to first
// order either way; what freezing drops is G h^2/6 a_grav per step in the velocity increment (k = 4e-14 s^-2, a_grav =
// 8.7 m/s^2: 6e-17 h^3 m/s, i.e. 1e-20 of the velocity per step at h = 0.1 s and 8e-18 at h = 1 s) and k v h^3 / 12 in
// the position (3e-14 m per step at h = 0.1 s) - both far below one ulp, like the anchor's own second-order term.  One
// evaluation (13 instructions) per step instead of four tidal products (40).  Rounds 2-3 evaluated A0 + G r per stage.
__device__ __forceinline__ V3 third_body_step(const Sun3& s3, V3 r, V3 v, double h2) {
    return tidal(s3, axpy(h2, v, r), s3.A0);
}
__device__ __forceinline__ void third_body_anchor(Sun3& s3, V3 sun, double mu, V3 rc) {
    const V3 a = third_body_exact(sun, mu, s3.k, rc);
    const V3 g = tidal(s3, rc, mk(0, 0, 0));
    s3.A0 = a - g;
}

// `base` is added to the result (the third body's constant part at the full-scenario levels: one FMA instead of a
// multiply and an add per component)
// `pre` (full-scenario single-wave kernels, first stage): |r|^2 and 1/|r| of this position are known already
template <int GRAV, int SPLIT, class Hot>
__device__ __forceinline__ V3 gravity(const Hot& c, V3 r, double tsim, V3 base, const Pre* pre = nullptr) {
    if constexpr (GRAV == BSK_GRAV_SH) {
        double sn, cs;
        sincos(c.planet_rate * tsim, &sn, &cs);
        const V3 pf = mk(fma(cs, r.x, sn * r.y), fma(cs, r.y, -sn * r.x), r.z);
        V3 af;
        if constexpr (SPLIT == 4) af = gravity_sh_dpp<false>(c, pf);
        else if constexpr (SPLIT == 5) af = gravity_sh_dpp<true>(c, pf);
        else af = gravity_sh(c, pf);
        return mk(fma(cs, af.x, fma(-sn, af.y, base.x)), fma(sn, af.x, fma(cs, af.y, base.y)), af.z + base.z);
    } else {
        double zz = r.z * r.z;
        double r2 = pre ? pre->r2 : fma(r.x, r.x, fma(r.y, r.y, zz));
        double ir = pre ? pre->ir : rsqrt_nr(r2);
        double ir2 = ir * ir;
        double ir3 = ir * ir2;
        double k0 = c.nmu * ir3;
        if constexpr (GRAV == BSK_GRAV_PM_J2) {
            double z2 = zz * ir2;               // (z/r)^2
            double kj = c.j2k * (ir3 * ir2);    // 1.5 J2 mu Re^2 / r^5
            double kxy = fma(kj, fma(5.0, z2, -1.0), k0);
            double kz = fma(-2.0, kj, kxy);     // k0 + kj (5 z2 - 3)
            return V3{fma(kxy, r.x, base.x), fma(kxy, r.y, base.y), fma(kz, r.z, base.z)};
        } else {
            return axpy(k0, r, base);
        }
    }
}

// --------------------------------------------------------------------------------------------
// equations of motion.
//   [I - sum Js g g^T] w' = L_ext - Gs (u + tau_f) - w x (I w + sum Js Om g)
//   Om_i' = (u_i + tau_f,i)/Js_i - g_i . w'
// The wheel torque tq_i = u_i + tau_f,i depends on the state only through sign(Om_i) (Coulomb
// friction), so the caller passes rhs0 = L_ext - sum tq_i g_i and tqj_i = tq_i / Js_i.
// Per-step environment handed to the equations of motion by the full-scenario levels (nothing at lower levels).
// The switches are WAVE-UNIFORM (a ballot over the lanes that need the term): every lane then evaluates the term
// and a lane that does not need it contributes exact zeros (density 0, empty thruster mask), so the per-lane
// results are those of a per-lane branch without its exec-mask bookkeeping.
struct Env {
    Sun3 s3;
    bool sun_on, drag_on, thr_on;
    double rho;              // density of this dyn tick, 0 below the skip threshold
    const ColdCfg* cold;
    KTab kt;                 // broadcast table (facet tables, 1/mass, ...)
    // thrusterDynamicEffector: current burst, on-time per thruster in half dyn steps (integers <= 2 * fsw_every <
    // 2^16, two per register: the slab keeps them as doubles), elapsed e2
    unsigned thr_lim2[BSK_MAX_THR / 2];
    int thr_max;      // max over thrusters of the limits (0 = no burst pending)
    int e2;
    // burst ticks only: which thrusters fire at the integrator times e2, e2 + 1, e2 + 2 (bit i = thruster i) and
    // the force / torque row of the first mask (ColdCfg::thr_tab); a stage whose mask differs reloads its own row
    int m0, m1, m2;
    V3 FB0, LB0;
    int facet_axis;   // generic-facet level only: ColdCfg::facet_axis
};

__device__ __forceinline__ int thr_limit(const Env& ev, int i) { return (int)((ev.thr_lim2[i >> 1] >> (16 * (i & 1))) & 0xFFFFu); }
__device__ __forceinline__ void thr_row(const ColdCfg* cc, int mask, V3& FB, V3& LB) {
    const double* row = cc->thr_tab[mask];
    FB = mk(row[0], row[1], row[2]);
    LB = mk(row[3], row[4], row[5]);
}
// per dyn tick with a burst pending somewhere in the wave: activity masks of the three integrator times
__device__ __forceinline__ void thr_masks(Env& ev) {
    int m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int i = 0; i < BSK_MAX_THR; ++i) {
        const int lim = thr_limit(ev, i);      // 0 = thruster not in the burst
        m0 |= (lim > 0 && ev.e2 <= lim) ? (1 << i) : 0;
        m1 |= (lim > 0 && ev.e2 + 1 <= lim) ? (1 << i) : 0;
        m2 |= (lim > 0 && ev.e2 + 2 <= lim) ? (1 << i) : 0;
    }
    ev.m0 = m0; ev.m1 = m1; ev.m2 = m2;
    thr_row(ev.cold, m0, ev.FB0, ev.LB0);
}
// thrust of the active thrusters at integrator stage `de2` (0, 1, 1, 2 half dyn steps after the tick started)
// PART (three-wave form): 0 = force and torque, PART_ROT = the torque only, PART_TRA = the force only
template <int PART = PART_ALL>
__device__ __forceinline__ void thrusters(const Env& ev, int de2, V3 sig, MrpRot rot, V3& aN, V3& LB) {
    V3 FB = ev.FB0;
    LB = ev.LB0;
    const int mk_ = de2 == 0 ? ev.m0 : (de2 == 1 ? ev.m1 : ev.m2);
    if (mk_ != ev.m0) thr_row(ev.cold, mk_, FB, LB);      // a pulse ends inside this dyn step
    if constexpr (PART == PART_ROT) { aN = mk(0, 0, 0); return; }
    const V3 u1 = cross(sig, FB), u2 = cross(sig, u1);
    const V3 FN = FB + rot.ka * u2 + rot.kb * u1;                                               // [BN]^T F_B
    aN = mk(mul_k<KC_IMASS>(ev.kt.c, FN.x), mul_k<KC_IMASS>(ev.kt.c, FN.y), mul_k<KC_IMASS>(ev.kt.c, FN.z));   // / m
}

// facet drag: F = -1/2 rho |v|^2 sum_i Cd_i A_i max(0, n_i . v_hat) v_hat,  L = sum_i r_i x F_i, with v
// the inertial velocity.  Only the projected-area sum S = sum c_i (n_i . v)+ and its moment
// Rc = sum c_i (n_i . v)+ r_i need the body frame; the force is along -v in ANY frame, so the
// inertial acceleration is -(1/2 rho S / m) v_N with no rotation back, and L_B = Rc x (-1/2 rho v_B).
// Accumulates the acceleration into aN and the torque into LB.
// PART_TRA / PART_ROT: only the projected-area sum and the acceleration / only its moment and the torque (each accumulator
// takes the same terms in the same order as in the whole form)
template <bool GENERIC, int PART = PART_ALL>
__device__ __forceinline__ void facet_drag(const Env& ev, V3 sig, MrpRot rot, V3 vN, V3& aN, V3& LB) {
    static_assert(PART == PART_ALL || !GENERIC, "the halves exist for the axis-aligned facet tables only");
    const double ka = rot.ka, kb = rot.kb;
    const V3 t1 = cross(sig, vN), t2 = cross(sig, t1);
    const V3 vB = vN + ka * t2 - kb * t1;                 // [BN] v
    // |v|^2 (n . v^)+ v^ = (n . v)+ v: the projected-area sums are homogeneous in v, so nothing is normalised
    const V3 vh = vB;
    double S = 0.0;
    V3 Rc = mk(0, 0, 0);
    // normals are +-e_k: the facets facing the flow on axis k are the +e_k ones when v_hat_k > 0, the
    // -e_k ones otherwise, and |x| sel(x > 0, p, m) = |x| (p + m)/2 + x (p - m)/2 needs no select
    // (the tables hold the half sums [0] and half differences [1]; |x| is a free source modifier)
    if (!GENERIC) {
        // facet centres on their own normal axes: 12 table values (row B of the broadcast table).  Four independent
        // accumulators, issued interleaved (a DPP FMA right behind the instruction that wrote one of its operands
        // is padded with s_nop); the area table carries 1/m, so Sm = S / m.
        const double tb = ev.kt.b;
        double rx = 0.0, ry = 0.0, rz = 0.0;
        if constexpr (PART == PART_TRA) {
            S = fmac_k<KB_FAC + 3>(S, tb, vh.x);
            S = fmac_k_abs<KB_FAC + 0>(S, tb, vh.x);
            S = fmac_k<KB_FAC + 4>(S, tb, vh.y);
            S = fmac_k_abs<KB_FAC + 1>(S, tb, vh.y);
            S = fmac_k<KB_FAC + 5>(S, tb, vh.z);
            S = fmac_k_abs<KB_FAC + 2>(S, tb, vh.z);
        } else if constexpr (PART == PART_ROT) {
            rx = fmac_k<KB_FAD + 3>(rx, tb, vh.x);     ry = fmac_k<KB_FAD + 4>(ry, tb, vh.y);     rz = fmac_k<KB_FAD + 5>(rz, tb, vh.z);
            rx = fmac_k_abs<KB_FAD + 0>(rx, tb, vh.x); ry = fmac_k_abs<KB_FAD + 1>(ry, tb, vh.y); rz = fmac_k_abs<KB_FAD + 2>(rz, tb, vh.z);
            Rc.x = rx; Rc.y = ry; Rc.z = rz;
        } else {
        S = fmac_k<KB_FAC + 3>(S, tb, vh.x);       rx = fmac_k<KB_FAD + 3>(rx, tb, vh.x);
        ry = fmac_k<KB_FAD + 4>(ry, tb, vh.y);     rz = fmac_k<KB_FAD + 5>(rz, tb, vh.z);
        S = fmac_k_abs<KB_FAC + 0>(S, tb, vh.x);   rx = fmac_k_abs<KB_FAD + 0>(rx, tb, vh.x);
        ry = fmac_k_abs<KB_FAD + 1>(ry, tb, vh.y); rz = fmac_k_abs<KB_FAD + 2>(rz, tb, vh.z);
        S = fmac_k<KB_FAC + 4>(S, tb, vh.y);       Rc.x = rx;
        S = fmac_k_abs<KB_FAC + 1>(S, tb, vh.y);   Rc.y = ry;
        S = fmac_k<KB_FAC + 5>(S, tb, vh.z);       Rc.z = rz;
        S = fmac_k_abs<KB_FAC + 2>(S, tb, vh.z);
        }
    } else if (ev.facet_axis == 1) {
        const ColdCfg* cc = ev.cold;
        const double vk[3] = {vh.x, vh.y, vh.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = fabs(vk[k]);
            S = fma(a, cc->fa_c[0][k], fma(vk[k], cc->fa_c[1][k], S));
            Rc = axpy(a, mk(cc->fa_r[0][k][0], cc->fa_r[0][k][1], cc->fa_r[0][k][2]),
                      axpy(vk[k], mk(cc->fa_r[1][k][0], cc->fa_r[1][k][1], cc->fa_r[1][k][2]), Rc));
        }
    } else {
        const ColdCfg* cc = ev.cold;
        for (int i = 0; i < cc->n_facets; ++i) {
            const double proj = fma(cc->facet_n[i][0], vh.x, fma(cc->facet_n[i][1], vh.y, cc->facet_n[i][2] * vh.z));
            if (proj > 0.0) {
                const double ci = cc->facet_acd[i] * proj;
                S += ci;
                Rc = axpy(ci, mk(cc->facet_r[i][0], cc->facet_r[i][1], cc->facet_r[i][2]), Rc);
            }
        }
    }
    const double kq = -0.5 * ev.rho;                      // 0 for a lane above the atmosphere
    if constexpr (PART != PART_TRA) LB = add_cross(LB, Rc, kq * vB);
    if constexpr (PART != PART_ROT) {
        if (!GENERIC) aN = axpy(S * kq, vN, aN);                        // S already carries 1/m
        else aN = axpy(mul_k<KC_IMASS>(ev.kt.c, S) * kq, vN, aN);
    }
}

// Integration state inside one RK4 step.  The hub sees the wheels only through their total
// momentum p = sum Js Om_i g_i (body frame), and with the wheel torque tq held over the step the
// momentum obeys      p' = T - W w',      T = sum tq_i g_i,  W = sum Js g g^T,
// which is LINEAR in w' with constant T: every RK4 stage value of p follows exactly from that stage's w,
//     p_i = p_0 + a_i T - W (w_i - w_0)          (a_i = 0, h/2, h/2, h),
// so p is not integrated at all.  The stage's total angular momentum is
//     H_i = I w_i + p_i = [I - W] w_i + c_i,     c_i = (p_0 + W w_0) + a_i T,
// with [I - W] the matrix whose inverse the back-substitution needs anyway:
//     [I - W] w' = L_ext - T - w x H.
// The four stages integrate (r, v, sigma, w) - 12 doubles - with `p` of the stage state holding c_i (three values per
// step: c_1, c_2 = c_3, c_4), and the individual wheel speeds follow EXACTLY (RK4 is linear in w') from
//     Om_i(t+h) = Om_i + h tq_i/Js_i - g_i . (w(t+h) - w(t)).
// That is the same map as RK4 on [.., Om_1..n], with 36 wheel operations per stage replaced by 3 (round 4; rounds 1-3
// integrated p beside w: 9 more per stage).
struct Core {
    V3 r, v, s, w, p;      // p: the stage's momentum offset c_i (above), not integrated
};

// THR: the step runs inside a thruster burst of some lane of the wave (the tick loop picks the instantiation, so the
// steps outside bursts - nearly all - carry no thruster code and no branch around it)
// DRAGM: 0 = the drag switch is tested where it is used (one wave-uniform branch per stage), 1 / 2 = the tick loop has
// picked the instantiation with / without drag, so that a stage is one branch-free scheduling region
// PART (three-wave form): PART_TRA evaluates r', v' only (x.s is then the OTHER wave's stage attitude), PART_ROT sigma', w', p'
// only (x.v is the other wave's stage velocity); every value either half produces is computed by the operations of the whole
// PH (the halves only): 1 = what does not need the other wave's value (r' and the gravity / Sun part of v'; sigma'),
// 2 = the rest (drag and thrust; w', p'), 0 = everything
template <int GRAV, int NRW, bool DIAG, int FEAT, int SPLIT, bool THR, int DRAGM = 0, int PART = PART_ALL, int PH = 0>
__device__ __forceinline__ void eom(const HotCfg<NRW, DIAG>& c, const Core& x, V3 rhs0, V3 a3, double tsim, const Env& ev,
                                    int de2, Core& d, const Pre* pre = nullptr) {
    static_assert(PH == 0 || PART != PART_ALL, "phases exist for the halves");
    if constexpr (PART != PART_ROT && PH != 2) d.r = x.v;
    if constexpr (is_full<FEAT>()) {
        // Sun third body: the step's constant (third_body_step), folded into the gravity FMAs
        if constexpr (PART != PART_ROT && PH != 2) d.v = gravity<GRAV, SPLIT>(c, x.r, tsim, a3, pre);
        if constexpr (PH != 1) {
            MrpRot rot{0.0, 0.0};       // rotation coefficients of this stage's attitude (drag, thrust)
            if constexpr (DRAGM != 2 || THR) rot = pre ? pre->rot : mrp_rot(x.s);
            if constexpr (DRAGM == 1) facet_drag<FEAT == FEAT_FULLG, PART>(ev, x.s, rot, x.v, d.v, rhs0);
            else if constexpr (DRAGM == 0) { if (BSK_LIKELY(ev.drag_on)) facet_drag<FEAT == FEAT_FULLG, PART>(ev, x.s, rot, x.v, d.v, rhs0); }
            if constexpr (THR) {
                V3 aN, LB;
                thrusters<PART>(ev, de2, x.s, rot, aN, LB);
                if constexpr (PART != PART_ROT) d.v = d.v + aN;
                if constexpr (PART != PART_TRA) rhs0 = rhs0 + LB;
            }
        }
    } else {
        if constexpr (PART != PART_ROT && PH != 2) d.v = gravity<GRAV, SPLIT>(c, x.r, tsim, a3);
    }
    if constexpr (PART == PART_TRA) return;
    if constexpr (PH != 2) {
    // sigma' = 1/4 [(1 - s^2) w + 2 s x w + 2 (s.w) s],  with hw = w/2:
    //        = (1 - s^2)/2 hw + s x hw + (s.hw) s
    V3 hw = 0.5 * x.w;
    double s2 = dot(x.s, x.s);
    double b = dot(x.s, hw);
    double a = fma(-0.5, s2, 0.5);
    d.s = V3{fma(a, hw.x, fma(b, x.s.x, fma(x.s.y, hw.z, -(x.s.z * hw.y)))),
             fma(a, hw.y, fma(b, x.s.y, fma(x.s.z, hw.x, -(x.s.x * hw.z)))),
             fma(a, hw.z, fma(b, x.s.z, fma(x.s.x, hw.y, -(x.s.y * hw.x))))};
    }
    if constexpr (PH == 1) return;
    V3 H;      // [I - W] w + c_i
    if constexpr (NRW > 0) {
        if constexpr (DIAG) H = V3{fma(c.Dm[0], x.w.x, x.p.x), fma(c.Dm[1], x.w.y, x.p.y), fma(c.Dm[2], x.w.z, x.p.z)};
        else H = mv(c.Dm, x.w) + x.p;
    } else {
        H = mv3<DIAG>(c.Dm, x.w);
    }
    d.w = mv3<DIAG>(c.Di, sub_cross(rhs0, x.w, H));
}

template <int NRW, int PART = PART_ALL>
__device__ __forceinline__ void core_axpy(double a, const Core& k, const Core& x, Core& o) {
    if constexpr (PART != PART_ROT) {
        o.r = axpy(a, k.r, x.r);
        o.v = axpy(a, k.v, x.v);
    }
    if constexpr (PART != PART_TRA) {
        o.s = axpy(a, k.s, x.s);
        o.w = axpy(a, k.w, x.w);
    }
}

// RK4 accumulator (12 doubles) staged in LDS (FEAT_LDSS): acc[f][lane], one 512-byte row per component and wave, every lane
// touches only its own column (conflict-free ds_read_b64 / ds_write_b64, no synchronisation).  volatile: the
// values must really leave the registers between the stages.
struct AccLds {
    double v[12][64];
};
typedef AccLds __attribute__((address_space(3))) * AccP;
template <int NRW>
__device__ __forceinline__ void acc_store(AccP A, int lane, const Core& a) {
    volatile double __attribute__((address_space(3)))* p = &A->v[0][lane];
    p[0 * 64] = a.r.x; p[1 * 64] = a.r.y; p[2 * 64] = a.r.z; p[3 * 64] = a.v.x; p[4 * 64] = a.v.y; p[5 * 64] = a.v.z;
    p[6 * 64] = a.s.x; p[7 * 64] = a.s.y; p[8 * 64] = a.s.z; p[9 * 64] = a.w.x; p[10 * 64] = a.w.y; p[11 * 64] = a.w.z;
}
template <int NRW>
__device__ __forceinline__ void acc_load(AccP A, int lane, Core& a) {
    const volatile double __attribute__((address_space(3)))* p = &A->v[0][lane];
    a.r = mk(p[0 * 64], p[1 * 64], p[2 * 64]); a.v = mk(p[3 * 64], p[4 * 64], p[5 * 64]);
    a.s = mk(p[6 * 64], p[7 * 64], p[8 * 64]); a.w = mk(p[9 * 64], p[10 * 64], p[11 * 64]);
}

// classic RK4, sequential accumulation x0 + h/6 k1 + h/3 k2 + h/3 k3 + h/6 k4, then the MRP
// shadow-set switch once per completed step.  Motor torque and Coulomb friction are evaluated
// from the wheel speeds at the start of the step and held through its four stages (the RW
// effector updates both once per dyn tick, outside the equations of motion).
template <int GRAV, int NRW, bool DIAG, int FEAT, int SPLIT, bool THR = false, int DRAGM = 0, class WV>
__device__ __forceinline__ void rk4_step(const HotCfg<NRW, DIAG>& c, const WV& wv, State<NRW>& x,
                                         const double* u, V3 lext, double t0, const Env& ev, AccP acc_lds = nullptr,
                                         const Pre* pre = nullptr, double* q2_out = nullptr) {
    Core y, k, yt, acc;
    y.r = x.r; y.v = x.v; y.s = x.s; y.w = x.w;
    y.p = mk(0, 0, 0);
    if constexpr (NRW > 0) y.p = mv3<DIAG>(c.W, y.w);      // W w0: the wheels' momentum goes on top (wv.head)
    V3 T = mk(0, 0, 0);
    double tqj[NRW > 0 ? NRW : 1], tq[NRW > 0 ? NRW : 1];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        // Coulomb friction -fc sign(Om), 0 at rest, as three fp64 instructions: fc sign(Om) = clamp(2^1000 Om, -fc, fc)
        // (exactly +-fc for every |Om| >= fc 2^-1000 - wheel speeds are 1e-10 rad/s and up - and exactly 0 at rest;
        // the sign-copy + compare + two selects it replaces took six)
        const double fs = fmin(fmax(x.Om[i] * 0x1p1000, -c.fc), c.fc);
        tq[i] = u[i] - fs;
    }
    if constexpr (NRW > 0) wv.head(tq, x.Om, T, y.p, tqj);      // y.p = c_1 = p_0 + W w_0
    const V3 rhs0 = lext - T;
    V3 c2 = y.p, c4 = y.p;                                       // c_2 = c_3 = c_1 + h/2 T,  c_4 = c_1 + h T
    if constexpr (NRW > 0) { c2 = axpy(c.h2, T, y.p); c4 = axpy(c.h, T, y.p); }
    V3 a3 = mk(0, 0, 0);
    if constexpr (is_full<FEAT>()) a3 = third_body_step(ev.s3, y.r, y.v, c.h2);
    constexpr bool LDSACC = FEAT == FEAT_LDSS;
    AccP A = nullptr;
    int lane = 0;
    if constexpr (LDSACC) { A = acc_lds; lane = (int)(threadIdx.x & 63u); }
    eom<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM>(c, y, rhs0, a3, t0, ev, 0, k, pre);
    core_axpy<NRW>(c.h6, k, y, acc);
    if constexpr (LDSACC) acc_store<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h2, k, y, yt);
    yt.p = c2;
    eom<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM>(c, yt, rhs0, a3, t0 + c.h2, ev, 1, k);
    if constexpr (LDSACC) acc_load<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h3, k, acc, acc);
    if constexpr (LDSACC) acc_store<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h2, k, y, yt);
    eom<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM>(c, yt, rhs0, a3, t0 + c.h2, ev, 1, k);
    if constexpr (LDSACC) acc_load<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h3, k, acc, acc);
    if constexpr (LDSACC) acc_store<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h, k, y, yt);
    yt.p = c4;
    eom<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM>(c, yt, rhs0, a3, t0 + c.h, ev, 2, k);
    if constexpr (LDSACC) acc_load<NRW>(A, lane, acc);
    core_axpy<NRW>(c.h6, k, acc, yt);
    const V3 dw = yt.w - y.w;
    double base[NRW > 0 ? NRW : 1];
    if constexpr (NRW > 0) {
        wv.bases(c.h, tq, tqj, x.Om, base);
        wv.tail(dw, base, x.Om);
    }
    x.r = yt.r; x.v = yt.v; x.s = yt.s; x.w = yt.w;
    double s2 = dot(x.s, x.s);
    if (BSK_UNLIKELY(s2 > 1.0)) {
        x.s = (-rcp_nr(s2)) * x.s;
        if (q2_out) s2 = dot(x.s, x.s);
    }
    if (q2_out) *q2_out = s2;        // |sigma|^2 of the step's attitude, for whoever rotates with it next (EnvTask)
}

// ---- three-wave form (SPLIT == 3): the RK4 step of one half of the spacecraft (exchange protocol: TriX).
// The translational half (r, v: gravity, Sun, drag force, thrust) needs the attitude of every stage for the drag and thrust
// rotations, the rotational half (sigma, omega, wheel momentum: kinematics, Euler, drag and thruster torque, wheels) the
// velocity.  Each wave integrates its own components with the operations of rk4_step, in its order per value; only WHEN a
// value is computed differs: a stage first does what is independent of the partner (the reads of the partner's value in
// flight meanwhile), the rotational wave publishes the NEXT stage's attitude as soon as the kinematics are through.
//   rotational:    prefetch v(i) | sigma' -> acc.s, sigma(i+1) -> publish | finish v(i) | drag torque, w', p' -> acc, w(i+1), p(i+1)
//   translational: prefetch sigma(i) | r', gravity, Sun -> acc.r, r(i+1) | finish sigma(i) | drag force -> acc.v, v(i+1) -> publish
// The fourth stage publishes the NEXT tick's first-stage value (attitude after the MRP switch; velocity together with the
// density at the new position); the tick loop publishes the first tick's before it starts.
// Rotational half: `v1` is the first-stage velocity (it came with the tick's density, which the tick loop needs to pick the
// instantiation), `tq`, `T`, `pw`, `tqj` the wheel terms of the tick (wv.head, evaluated by the tick loop while v1 was in
// flight).  Translational half: those are unused; `density` evaluates the atmosphere, rho_next returns the next tick's.
template <int GRAV, int NRW, bool DIAG, int FEAT, int SPLIT, bool THR, int DRAGM, int PART, class WV, class XC, class RHO>
__device__ __forceinline__ void rk4_step_part(const HotCfg<NRW, DIAG>& c, const WV& wv, State<NRW>& x, V3 lext, double t0,
                                              const Env& ev, XC& xc, V3 v1, const double* tq, V3 T, V3 pw, const double* tqj,
                                              RHO&& density, double& rho_next) {
    static_assert(PART == PART_ROT || PART == PART_TRA, "one half");
    // a tick without drag and outside thruster bursts couples the halves through nothing: only the tick-boundary values
    // are exchanged then (they carry the next tick's density, and they keep either wave within a tick of the other)
    constexpr bool COUPLED = THR || DRAGM != 2;
    Core y, k, cur, nxt, acc;
    double none;
    TriPend f;
    V3 a3 = mk(0, 0, 0);
    if constexpr (PART == PART_TRA) a3 = third_body_step(ev.s3, x.r, x.v, c.h2);
#define BSK_EOM(PH, Z, TS, DE) eom<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM, PART, PH>(c, Z, rhs0, a3, TS, ev, DE, k)
    if constexpr (PART == PART_ROT) {
        const V3 rhs0 = lext - T;
        y.s = x.s; y.w = x.w; y.p = pw; y.v = v1;               // pw = c_1 = p_0 + W w_0 (the tick loop's wv.head)
        V3 c2 = pw, c4 = pw;
        if constexpr (NRW > 0) { c2 = axpy(c.h2, T, pw); c4 = axpy(c.h, T, pw); }
        // stage 1 (its velocity came with the tick's density: consumed by the tick loop)
        BSK_EOM(1, y, t0, 0);
        acc.s = axpy(c.h6, k.s, y.s);
        nxt.s = axpy(c.h2, k.s, y.s);
        if constexpr (COUPLED) xc.template publish<false>(nxt.s, 0.0);
        BSK_EOM(2, y, t0, 0);
        acc.w = axpy(c.h6, k.w, y.w);
        nxt.w = axpy(c.h2, k.w, y.w);
        nxt.p = c2;
        cur = nxt;
        // stage 2
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h2, 1);
        acc.s = axpy(c.h3, k.s, acc.s);
        nxt.s = axpy(c.h2, k.s, y.s);
        if constexpr (COUPLED) xc.template publish<false>(nxt.s, 0.0);
        if constexpr (COUPLED) cur.v = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h2, 1);
        acc.w = axpy(c.h3, k.w, acc.w);
        nxt.w = axpy(c.h2, k.w, y.w);
        cur = nxt;
        // stage 3
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h2, 1);
        acc.s = axpy(c.h3, k.s, acc.s);
        nxt.s = axpy(c.h, k.s, y.s);
        if constexpr (COUPLED) xc.template publish<false>(nxt.s, 0.0);
        if constexpr (COUPLED) cur.v = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h2, 1);
        acc.w = axpy(c.h3, k.w, acc.w);
        nxt.w = axpy(c.h, k.w, y.w);
        nxt.p = c4;
        cur = nxt;
        // stage 4: the step's attitude, switched to the inner MRP set where needed, is the next tick's first-stage value
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h, 2);
        x.s = axpy(c.h6, k.s, acc.s);
        const double s2 = dot(x.s, x.s);
        if (BSK_UNLIKELY(s2 > 1.0)) x.s = (-rcp_nr(s2)) * x.s;
        xc.template publish<false>(x.s, 0.0);
        if constexpr (COUPLED) cur.v = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h, 2);
        nxt.w = axpy(c.h6, k.w, acc.w);
        const V3 dw = nxt.w - y.w;
        double base[NRW > 0 ? NRW : 1];
        if constexpr (NRW > 0) {
            wv.bases(c.h, tq, tqj, x.Om, base);
            wv.tail(dw, base, x.Om);
        }
        x.w = nxt.w;
    } else {
        const V3 rhs0 = mk(0, 0, 0);
        y.r = x.r; y.v = x.v;
        // stage 1
        f = xc.template prefetch<false>();
        BSK_EOM(1, y, t0, 0);
        acc.r = axpy(c.h6, k.r, y.r);
        nxt.r = axpy(c.h2, k.r, y.r);
        y.s = xc.template finish<false>(f, none);
        BSK_EOM(2, y, t0, 0);
        acc.v = axpy(c.h6, k.v, y.v);
        nxt.v = axpy(c.h2, k.v, y.v);
        if constexpr (COUPLED) xc.template publish<false>(nxt.v, 0.0);
        cur = nxt;
        // stage 2
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h2, 1);
        acc.r = axpy(c.h3, k.r, acc.r);
        nxt.r = axpy(c.h2, k.r, y.r);
        if constexpr (COUPLED) cur.s = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h2, 1);
        acc.v = axpy(c.h3, k.v, acc.v);
        nxt.v = axpy(c.h2, k.v, y.v);
        if constexpr (COUPLED) xc.template publish<false>(nxt.v, 0.0);
        cur = nxt;
        // stage 3
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h2, 1);
        acc.r = axpy(c.h3, k.r, acc.r);
        nxt.r = axpy(c.h, k.r, y.r);
        if constexpr (COUPLED) cur.s = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h2, 1);
        acc.v = axpy(c.h3, k.v, acc.v);
        nxt.v = axpy(c.h, k.v, y.v);
        if constexpr (COUPLED) xc.template publish<false>(nxt.v, 0.0);
        cur = nxt;
        // stage 4: the step's velocity goes out together with the density at the step's position
        if constexpr (COUPLED) f = xc.template prefetch<false>();
        BSK_EOM(1, cur, t0 + c.h, 2);
        x.r = axpy(c.h6, k.r, acc.r);
        rho_next = density(x.r);     // (a serial chain through rsq and exp: here, beside the reads in flight, not at the tick's end)
        if constexpr (COUPLED) cur.s = xc.template finish<false>(f, none);
        BSK_EOM(2, cur, t0 + c.h, 2);
        x.v = axpy(c.h6, k.v, acc.v);
        xc.template publish<true>(x.v, rho_next);
    }
#undef BSK_EOM
}

// --------------------------------------------------------------------------------------------
// attitude kinematics helpers (FSW chain / observation; run at 1/10 of the RK4 rate)
__device__ __forceinline__ void mrp2c(V3 q, double* C) {
    double q2 = dot(q, q), op = 1.0 + q2, id = rcp_nr(op * op);
    double a = 8.0 * id, b = 4.0 * (1.0 - q2) * id;
    // t~^2 = q q^T - q2 I
    C[0] = fma(a, q.x * q.x - q2, 1.0);
    C[4] = fma(a, q.y * q.y - q2, 1.0);
    C[8] = fma(a, q.z * q.z - q2, 1.0);
    double xy = a * q.x * q.y, xz = a * q.x * q.z, yz = a * q.y * q.z;
    C[1] = fma(b, q.z, xy);
    C[3] = fma(-b, q.z, xy);
    C[2] = fma(-b, q.y, xz);
    C[6] = fma(b, q.y, xz);
    C[5] = fma(b, q.x, yz);
    C[7] = fma(-b, q.x, yz);
}

// DCM -> MRP through Euler parameters (Sheppard), b0 >= 0
__device__ __forceinline__ V3 c2mrp(const double* C) {
    double tr = C[0] + C[4] + C[8];
    double b20 = 0.25 * (1.0 + tr), b21 = 0.25 * (1.0 + 2.0 * C[0] - tr), b22 = 0.25 * (1.0 + 2.0 * C[4] - tr),
           b23 = 0.25 * (1.0 + 2.0 * C[8] - tr);
    int i = 0;
    double mx = b20;
    if (b21 > mx) { mx = b21; i = 1; }
    if (b22 > mx) { mx = b22; i = 2; }
    if (b23 > mx) { mx = b23; i = 3; }
    double ip = rsqrt_nr(mx), p = mx * ip, q4 = 0.25 * ip;   // mx >= 1/4 always
    // numerators of the four Sheppard cases, selected without divergent control flow
    double d0 = C[5] - C[7], d1 = C[6] - C[2], d2 = C[1] - C[3];
    double s0 = C[1] + C[3], s1 = C[6] + C[2], s2 = C[5] + C[7];
    double b0 = (i == 0) ? p : q4 * ((i == 1) ? d0 : (i == 2) ? d1 : d2);
    double b1 = (i == 1) ? p : q4 * ((i == 0) ? d0 : (i == 2) ? s0 : s1);
    double b2 = (i == 2) ? p : q4 * ((i == 0) ? d1 : (i == 1) ? s0 : s2);
    double b3 = (i == 3) ? p : q4 * ((i == 0) ? d2 : (i == 1) ? s1 : s2);
    if (b0 < 0.0) { b0 = -b0; b1 = -b1; b2 = -b2; b3 = -b3; }
    double id = rcp_nr(1.0 + b0);
    return V3{b1 * id, b2 * id, b3 * id};
}

// q1 (-) q2 with the near-singular guard and the map to the inner set
__device__ __forceinline__ V3 submrp(V3 q1, V3 q2) {
    double d1 = dot(q1, q1), d2 = dot(q2, q2);
    double den = 1.0 + d1 * d2 + 2.0 * dot(q1, q2);
    if (fabs(den) < 0.1) {
        q1 = (-rcp_nr(d1)) * q1;
        d1 = dot(q1, q1);
        den = 1.0 + d1 * d2 + 2.0 * dot(q1, q2);
    }
    V3 t = cross(q1, q2);
    double id = rcp_nr(den);
    V3 q = id * ((1.0 - d2) * q1 - (1.0 - d1) * q2 + 2.0 * t);
    double m = dot(q, q);
    if (m > 1.0) q = (-rcp_nr(m)) * q;
    return q;
}

struct Guid {
    V3 sigma_BR, omega_BR_B, omega_RN_B, domega_RN_B;
};

// hillPoint: the orbit frame's attitude, rate and acceleration from (r, v).
// ZNAV: a navigation message nobody has written yet may be among the lanes (all zeros: the FSW tick at t = 0 with
// bsk_config.nav_lag) - hillPoint's unit vectors normalise to zero, the zero DCM maps to the zero MRP and its radius guard
// zeroes the rates.  That instantiation carries 34 selects the chain of every other tick does without.
template <bool ZNAV>
__device__ __forceinline__ void hill_point(V3 r, V3 v, V3& sRN, V3& wRN_N, V3& dwRN_N) {
    bool znav = false;
    if constexpr (ZNAV) znav = dot(r, r) == 0.0;
    const V3 xr = znav ? mk(1, 0, 0) : r, xv = znav ? mk(0, 1, 0) : v;   // keeps the arithmetic finite
    double ir = rsqrt_nr(dot(xr, xr));
    V3 h = cross(xr, xv);
    double h2 = dot(h, h), ih = rsqrt_nr(h2), hm = h2 * ih;
    V3 e_r = ir * xr, e_h = ih * h, e_t = cross(e_h, e_r);
    double C[9] = {e_r.x, e_r.y, e_r.z, e_t.x, e_t.y, e_t.z, e_h.x, e_h.y, e_h.z};
    sRN = c2mrp(C);
    double dfdt = hm * ir * ir;
    double ddfdt2 = -2.0 * dot(xv, e_r) * ir * dfdt;
    wRN_N = dfdt * e_h;
    dwRN_N = ddfdt2 * e_h;
    if (znav) { sRN = mk(0, 0, 0); wRN_N = mk(0, 0, 0); dwRN_N = mk(0, 0, 0); }
}

// hillPoint | inertial3D  ->  attTrackingError
template <int NRW>
__device__ __forceinline__ Guid guidance(const double* __restrict__ sigma_R0N, const State<NRW>& x, int action) {
    V3 sRN, wRN_N, dwRN_N;
    // some lane of the wave holds the unwritten message (one ballot per FSW tick; a flag handed down from the tick loop
    // would occupy a scalar register pair across it)
    const bool maybe_znav = __builtin_amdgcn_ballot_w64(dot(x.r, x.r) == 0.0) != 0;
    if (action == 0) {
        if (BSK_UNLIKELY(maybe_znav)) hill_point<true>(x.r, x.v, sRN, wRN_N, dwRN_N);
        else hill_point<false>(x.r, x.v, sRN, wRN_N, dwRN_N);
    } else {
        sRN = mk(sigma_R0N[0], sigma_R0N[1], sigma_R0N[2]);
        wRN_N = mk(0, 0, 0);
        dwRN_N = mk(0, 0, 0);
    }
    Guid g;
    g.sigma_BR = submrp(x.s, sRN);
    double BN[9];
    mrp2c(x.s, BN);
    g.omega_RN_B = mv(BN, wRN_N);
    g.domega_RN_B = mv(BN, dwRN_N);
    g.omega_BR_B = x.w - g.omega_RN_B;
    return g;
}

// The FSW chain's constants: the head of the cold block (28 doubles), fetched in ONE batch when an FSW tick starts.
// Read where they are used they cost the chain two dependent memory round trips (sigma_R0N inside the guidance
// branch, the gains after it), and a wave that is alone on its SIMD waits each of them out.
struct FswCfg {
    double inertia[9];
    double map[BSK_MAX_RW][3];
    double u_max, u_min, K, P;
    double sigma_R0N[3];
};
static_assert(offsetof(ColdCfg, sigma_R0N) + sizeof(double[3]) == sizeof(FswCfg) && offsetof(ColdCfg, map) == offsetof(FswCfg, map) &&
              offsetof(ColdCfg, K) == offsetof(FswCfg, K), "FswCfg mirrors the head of ColdCfg");
__device__ __forceinline__ FswCfg load_fsw(const ColdCfg* __restrict__ c) {
    FswCfg f;
    const double* __restrict__ src = (const double*)c;
    double* dst = (double*)&f;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(FswCfg) / sizeof(double)); ++k) dst[k] = src[k];
    return f;
}

// MRP_Feedback -> rwMotorTorque -> wheel saturation / dead-band
template <int NRW>
__device__ __forceinline__ void control(const FswCfg& c, const Guid& g, double* u) {
    V3 wBN = g.omega_BR_B + g.omega_RN_B;
    V3 Lr = c.K * g.sigma_BR + c.P * g.omega_BR_B;
    Lr = Lr - cross(g.omega_RN_B, mv(c.inertia, wBN));
    Lr = Lr + mv(c.inertia, cross(wBN, g.omega_RN_B) - g.domega_RN_B);
    // module output is -Lr (torque on the body); wheels need u_s = -map * (-Lr) = map * Lr
    const double u_max = c.u_max, u_min = c.u_min;
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        double us = fma(c.map[i][0], Lr.x, fma(c.map[i][1], Lr.y, c.map[i][2] * Lr.z));
        if (u_max > 0.0) us = fmin(fmax(us, -u_max), u_max);
        if (fabs(us) < u_min) us = 0.0;
        u[i] = us;
    }
}

// rwDesatTask at an FSW tick in mode 2 (reference leoPowerAttitudeSimulator.py:452-478, 488-490,
// 574-588): thrMomentumManagement (one request per mode entry) -> thrForceMapping (on-pulsing
// minimum-norm impulses, smallest subtracted) -> thrMomentumDumping (bursts of at most one control
// period every thr_max_counter+1 periods, pulses below thrMinFireTime dropped, stretched to the
// thruster's MinOnTime).  thr_lim is the new burst in half dyn steps.
template <int NRW>
__device__ __forceinline__ void desat_tick(const ColdCfg* __restrict__ cc, const double* Om, bool first, double Tc,
                                           double two_over_dt, int fsw_every, double* __restrict__ rem_base,
                                           int64_t S, uint32_t bo, unsigned* lim2_new, bool& fired, int& thr_cnt) {
    // on-time still owed per thruster lives in the state slab and is touched only here (one request per mode entry,
    // one burst every thr_max_counter + 1 control periods): nothing of it is held across the RK4 loop
    double thr_rem[BSK_MAX_THR];
    if (first) {
        V3 hs = mk(0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRW; ++i) hs = axpy(cc->js[i] * Om[i], mk(cc->gs[i][0], cc->gs[i][1], cc->gs[i][2]), hs);
        const double hm = sqrt(dot(hs, hs));
        V3 dH = mk(0, 0, 0);
        if (hm > cc->hs_min) dH = (-(hm - cc->hs_min) / hm) * hs;
        double F[BSK_MAX_THR], fmin = 0.0;
#pragma unroll
        for (int i = 0; i < BSK_MAX_THR; ++i) {
            F[i] = (i < cc->n_thr) ? fma(cc->thr_map[i][0], dH.x, fma(cc->thr_map[i][1], dH.y, cc->thr_map[i][2] * dH.z)) : 0.0;
            if (i == 0 || (i < cc->n_thr && F[i] < fmin)) fmin = F[i];
        }
#pragma unroll
        for (int i = 0; i < BSK_MAX_THR; ++i) thr_rem[i] = (i < cc->n_thr) ? (F[i] - fmin) * cc->inv_max_thrust : 0.0;
        thr_cnt = 0;
    } else if (thr_cnt <= 0) {
#pragma unroll
        for (int i = 0; i < BSK_MAX_THR; ++i) thr_rem[i] = ldf(rem_base + (int64_t)i * S, bo);
    }
    if (thr_cnt <= 0) {
        unsigned lim[BSK_MAX_THR];
#pragma unroll
        for (int i = 0; i < BSK_MAX_THR; ++i) {
            lim[i] = 0u;
            if (i >= cc->n_thr) continue;
            double on = fmin(thr_rem[i], Tc);
            if (on < cc->thr_min_fire_time) {
                thr_rem[i] = 0.0;
            } else {
                thr_rem[i] -= on;
                lim[i] = (on >= Tc) ? 2u * (unsigned)fsw_every : (unsigned)floor(fmax(on, cc->thr_min_on_time) * two_over_dt);
            }
        }
#pragma unroll
        // the on-time command message: the thruster set latches it (and the burst starts) when the dynamics task runs
        for (int i = 0; i < BSK_MAX_THR / 2; ++i) lim2_new[i] = lim[2 * i] | (lim[2 * i + 1] << 16);
        fired = true;
        thr_cnt = cc->thr_max_counter;
#pragma unroll
        for (int i = 0; i < BSK_MAX_THR; ++i)
            *(gptr<double>)((gptr<char>)(gptr<double>)(rem_base + (int64_t)i * S) + bo) = thr_rem[i];
    } else {
        thr_cnt -= 1;
    }
}

// Sum over the 64 lanes of a wave in the order of the xor butterfly v += v[lane ^ off], off = 32, 16, ..., 1 - as far as lane 0 is
// concerned, which is the only lane whose result is used: at every level the lanes below `off` add the value `off` lanes up
// (lane ^ off = lane + off there, and the addition commutes bit for bit).  The two upper levels cross 16-lane rows (ds_bpermute),
// the four lower ones stay inside a row: DPP row shifts, no trip through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_pull(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_down(v, 32, 64);
    v += __shfl_down(v, 16, 64);
    v += dpp_pull<0x108>(v);      // row_shl:8  lane i <- lane i + 8
    v += dpp_pull<0x104>(v);      // row_shl:4
    v += dpp_pull<0x102>(v);      // row_shl:2
    v += dpp_pull<0x101>(v);      // row_shl:1
    return v;                     // (lane 0)
}

}  // namespace bsk
