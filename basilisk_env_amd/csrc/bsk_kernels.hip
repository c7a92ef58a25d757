// bsk_kernels.hip — gfx950 kernels of the batched propagator.
//
// step_kernel<GRAV, NRW, DIAG, FEAT, SPLIT>: one spacecraft per lane, 64-lane wavefronts.
//   GRAV  point mass / + J2 / spherical harmonics       NRW   0, 3 or 4 reaction wheels
//   DIAG  inertia and back-substitution matrix diagonal  FEAT  0 bare, 1 + power, 2 full scenario
//   SPLIT form of the harmonics evaluation (bsk_device.hpp: 1 scalar stream, 4 / 5 DPP broadcast with one / two waves)
//   HBM layout: structure-of-arrays fp64, field f of env i at st[f*stride + i]; a wave reads 512
//   contiguous bytes per field (global_load_dwordx2 per lane, fully coalesced), state lives in
//   VGPRs for all `substeps` RK4 steps, and is written back once.  The done flags are reduced per
//   wavefront (__ballot: one 64-bit mask, one store per wave); the batch's reward sum is formed by stats_kernel
//   from the reward buffer when somebody asks for it — no atomics, bitwise reproducible.
// (bsk_aux.hip: the small kernels around it - batch scalars, on-device IC sampler, resets.)
//
// Replaces run_sim + reward/done logic for N spacecraft:
//   reference basilisk_env/simulators/leoPowerAttitudeSimulator.py:535-644
//   reference basilisk_env/envs/leoPowerAttitudeEnvironment.py:98-127,161-170
#include "bsk_device.hpp"
#include "bsk_launch.hpp"

#include <atomic>
#include <cstddef>
#define __COMMA__ ,

namespace bsk {

// a wave-uniform condition as an integer in a scalar register
__device__ __forceinline__ int uni(bool b) { return __builtin_amdgcn_readfirstlane(b ? 1 : 0); }
// a wave-uniform double as a scalar value of its own: a kernel argument that arrived in a 16-dword scalar load is otherwise
// kept - and, when scalar registers run out, spilled and reloaded - as part of that whole tuple
__device__ __forceinline__ double own_scalar(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// ... "some lane of the wave": compare + s_cmp_lg_u64 + s_cselect_b32 (through uni() the compiler materialises the ballot as
// a lane value and reads it back: seven instructions)
__device__ __forceinline__ int uni_any(bool lane_cond) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(lane_cond);
    int r;
    asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return r;
}

__device__ __forceinline__ int wave_min_uniform(int v) {
    // every lane holds the same value unless a masked reset staggered the FSW phases inside this wave: one ballot
    // settles the common case, the shuffle tree (six trips through the LDS crossbar) only runs when lanes differ
    const int first = __builtin_amdgcn_readfirstlane(v);
    if (__builtin_amdgcn_ballot_w64(v != first) == 0) return first;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

constexpr int LDSS_WAVES = 3;   // waves per SIMD the LDS-scratch level is built for (168 VGPRs)
// SPLIT = 5 (harmonics only): a 256-thread workgroup carries 2 x 64 spacecraft; the two waves of a pair run the
// cheap RK4 redundantly (bit-identical), each walks half of the Pines entries and they exchange partial
// sums through LDS: twice the waves per SIMD at the same batch size, so one wave's loads and scalar
// instructions overlap with the other's FMAs.  Only wave 0 stores.
template <int GRAV, int NRW, bool DIAG, int FEAT, int SPLIT>
__global__ __launch_bounds__(256, (SPLIT == 5 || SPLIT == 4 || SPLIT == 2 || SPLIT == 3) ? 2 : (FEAT == FEAT_LDSS ? LDSS_WAVES : 1)) void step_kernel(const StepArgs<NRW, DIAG> a) {
    const HotCfg<NRW, DIAG>& c = a.hot;
    const ColdCfg* __restrict__ cold = a.cold;
    const probe::Stamp t_kernel = probe::stamp<probe::PAIR_TIME>();
    // SPLIT == 5: waves 0/1 of the workgroup carry spacecraft group 0 (halves 0/1 of the walk), waves 2/3 group 1
    // SPLIT == 2 (pair form, bsk_device.hpp: PairLds): a 128-thread workgroup = the dynamics wave and the FSW + environment
    // wave of the SAME 64 spacecraft
    // SPLIT == 3 (three-wave form, bsk_device.hpp: TriX): a 192-thread workgroup = the rotational wave, the FSW + environment
    // wave and the translational wave of the same 64 spacecraft
    constexpr bool TRI = SPLIT == 3;
    constexpr bool PAIR = SPLIT == 2 || TRI;
    const int gid = (SPLIT == 5) ? (int)(blockIdx.x * 128 + (threadIdx.x >> 7) * 64 + (threadIdx.x & 63))
                  : PAIR ? (int)(blockIdx.x * 64 + (threadIdx.x & 63))
                         : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool valid = gid < a.n;
    const int i = valid ? gid : a.n - 1;  // tail lanes shadow the last env; their stores are masked
    const int64_t S = a.stride;
    const double* __restrict__ st = a.st;
    const uint32_t bo = (uint32_t)i * 8u;   // per-lane byte offset inside every field row (n < 2^28)
#define FLD(f) (st + (int64_t)(f) * S)

    // every load of the launch is issued here, before any use
    State<NRW> x;
    x.r = mk(ldf(FLD(BSK_F_R + 0), bo), ldf(FLD(BSK_F_R + 1), bo), ldf(FLD(BSK_F_R + 2), bo));
    x.v = mk(ldf(FLD(BSK_F_V + 0), bo), ldf(FLD(BSK_F_V + 1), bo), ldf(FLD(BSK_F_V + 2), bo));
    x.s = mk(ldf(FLD(BSK_F_SIGMA + 0), bo), ldf(FLD(BSK_F_SIGMA + 1), bo), ldf(FLD(BSK_F_SIGMA + 2), bo));
    x.w = mk(ldf(FLD(BSK_F_OMEGA + 0), bo), ldf(FLD(BSK_F_OMEGA + 1), bo), ldf(FLD(BSK_F_OMEGA + 2), bo));
#pragma unroll
    for (int k = 0; k < NRW; ++k) x.Om[k] = ldf(FLD(BSK_NF_BASE + k), bo);
    constexpr int TAIL = BSK_NF_BASE + NRW;
    const V3 lext = mk(ldf(FLD(TAIL + BSK_T_LEXT + 0), bo), ldf(FLD(TAIL + BSK_T_LEXT + 1), bo),
                       ldf(FLD(TAIL + BSK_T_LEXT + 2), bo));
    double charge = 1.0;     // (bare levels with StepArgs::static_charge: never read - any non-zero value)
    if (FEAT >= FEAT_POWER || BSK_UNLIKELY(a.static_charge == 0)) charge = ldf(FLD(TAIL + BSK_T_CHARGE), bo);
    const int2 cnt = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(a.cnt) + bo);  // {steps | phase << 20, ticks}
    // (int32 actions, or the low words of int64 ones - torch's argmax output - read in place: a.act_shift)
    const int action = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.act) + (bo >> a.act_shift));
    // |sigma_BR| of the att_guidance message the last FSW tick wrote (obs[0] with bsk_config.nav_lag)
    double sbr = ldf(FLD(TAIL + BSK_T_SBR), bo);
    double u[NRW > 0 ? NRW : 1], up[NRW > 0 ? NRW : 1];
#pragma unroll
    for (int k = 0; k < (NRW > 0 ? NRW : 1); ++k) up[k] = 0.0;
    // the held motor torque only matters when this launch starts between two FSW ticks; it is
    // loaded unconditionally so that all loads are in flight at once
#pragma unroll
    for (int k = 0; k < NRW; ++k) u[k] = ldf(FLD(TAIL + BSK_T_UCMD + k), bo);
#undef FLD
    // device-resident episode statistics (BSK_FLAG_EPISODE_STATS): the running return travels with the state's loads
    double ep_ret = 0.0;
    if (BSK_UNLIKELY(a.ep_return != nullptr)) ep_ret = ldf(a.ep_return, bo);
    // The loop's first scalar constants, fetched while the state's loads are in flight: left to the compiler their
    // scalar loads sit behind the wait for the state - one more round trip on a K = 1 launch's critical path.
    asm volatile("" ::"s"(c.fsw_every), "s"(c.h), "s"(c.h2), "s"(c.h3), "s"(c.h6), "s"(c.nmu), "s"(c.j2k), "s"(c.Dm[0]), "s"(c.Dm[1]));

    // desaturation state (full scenario with BSK_FLAG_DESAT)
    constexpr bool FULL = is_full<FEAT>();
    int thr_t0 = 0, thr_cnt = 0;
    bool desat = false;
    if constexpr (FULL) desat = a.extra.desat != 0;

    const int steps0 = cnt.x & 0xFFFFF;
    int phase = cnt.x >> 20;
    bool fsw_ran = false;

    // Outer loop over FSW periods, inner loop of pure RK4 steps: the 1 Hz FSW chain (and the
    // SGPRs its constants need) stays out of the inner loop, which holds only HotCfg.
    const int fsw_every = c.fsw_every;
    const int substeps = a.substeps;
    // wheel geometry: parked in VGPRs, except at the full-scenario levels where it comes through the DPP broadcast
    // table (bsk_device.hpp: KTab)
    // (the LDS-scratch level is after a third wave per SIMD: it takes the 36 registers of the wheel geometry too)
    constexpr bool WDPP = FULL || (FEAT == FEAT_LDSS && NRW > 0);
    std::conditional_t<WDPP, WheelDpp<NRW, FULL>, WheelV<NRW>> wv;
    const int substeps_eff = substeps;
    int j = 0;
    int tick = cnt.y;
    bool tri_failed = false;
    double shadow = 1.0;
    SunGeom sg;
    constexpr bool POWER = FEAT >= FEAT_POWER;
    extern __shared__ __align__(16) unsigned char lds_dyn[];
    LdsP L = nullptr;
    const int lane = (int)(threadIdx.x & 63u);
    if constexpr (POWER) {
        sg = sun_setup(a.power, (double)tick * c.h);
        if constexpr (!PAIR) {
            L = (LdsP)lds_dyn + (threadIdx.x >> 6);
            L->sun[0][lane] = sg.sun.x; L->sun[1][lane] = sg.sun.y; L->sun[2][lane] = sg.sun.z;
            if (lane == 0) L->qcount = 0;
        }
    }
    AccP accp = nullptr;
    if constexpr (FEAT == FEAT_LDSS) accp = (AccP)lds_dyn + (threadIdx.x >> 6);
    KTab kt;
    if constexpr (FULL) {
        // lane l of every 16-lane row fetches entry l & 15 of the three table rows (three coalesced loads)
        const int l16 = lane & 15;
        kt.a = cold->kt[l16]; kt.b = cold->kt[16 + l16]; kt.c = cold->kt[32 + l16]; kt.e = cold->kt[64 + l16];
        wv.ta = kt.a; wv.tb = kt.b; wv.td = cold->kt[48 + l16];
    } else if constexpr (WDPP) {
        const int l16 = lane & 15;
        wv.ta = cold->kt[l16]; wv.tb = cold->kt[16 + l16]; wv.td = cold->kt[48 + l16];
    } else {
        wv.load(c);
    }
    Env ev;
    if constexpr (FULL) {
        ev.cold = cold;
        ev.kt = kt;
        ev.sun_on = a.extra.mu_sun != 0.0;
        ev.drag_on = false;
        ev.rho = 0.0;
        ev.s3.sh = sg.ism * sg.sun;
        ev.s3.k = a.extra.mu_sun * sg.ism * sg.ism * sg.ism;
        ev.s3.A0 = mk(0, 0, 0);
        ev.thr_on = false;
        ev.e2 = 0;
        ev.m0 = ev.m1 = ev.m2 = 0;
        ev.FB0 = mk(0, 0, 0);
        ev.LB0 = mk(0, 0, 0);
        ev.facet_axis = cold->facet_axis;
#pragma unroll
        for (int k = 0; k < BSK_MAX_THR / 2; ++k) ev.thr_lim2[k] = 0u;
        ev.thr_max = 0;
        if (desat) {
            // the current burst's limits are small integers kept as doubles in the slab: packed two per register here
#pragma unroll
            for (int k = 0; k < BSK_MAX_THR; ++k) {
                const unsigned lim = (unsigned)ldf(st + (int64_t)(TAIL + BSK_T_THR_LIM + k) * S, bo);
                ev.thr_lim2[k >> 1] |= lim << (16 * (k & 1));
                ev.thr_max = max(ev.thr_max, (int)lim);
            }
            thr_t0 = (int)ldf(st + (int64_t)(TAIL + BSK_T_THR_T0) * S, bo);
            thr_cnt = (int)ldf(st + (int64_t)(TAIL + BSK_T_THR_CNT) * S, bo);
        }
    }
    // what the previous tick's EnvTask knows about this tick's initial state (bsk_device.hpp: Pre): here of the loaded state
    Pre pre{0.0, 0.0, MrpRot{0.0, 0.0}};
    if constexpr (FULL && !PAIR) { pre.r2 = dot(x.r, x.r); pre.rot = mrp_rot(x.s); }
    Atmo atm{0.0, 0.0};   // exponentialAtmosphere along the trajectory (bsk_device.hpp): anchored per chunk of ticks, advanced per tick
    bool first_fsw = true;
    bool drag_cfg = false;
    if constexpr (FULL) drag_cfg = a.extra.base_density != 0.0;
    const int drag_cfg_u = uni(drag_cfg);
    // ---- FSW task timing (bsk_config.nav_lag) -------------------------------------------------------------------
    // The reference creates its FSW tasks with priorities 100 / 50 and its dynamics tasks with the default
    // (...Simulator.py:383-386, :101-103); Basilisk runs higher priorities first at equal time.  So the FSW tick of
    // time k F dt executes BEFORE the dynamics task integrates to that time: on the navigation / wheel-speed messages
    // of one integrator step earlier, and the effectors latch its commands when the dynamics task runs, i.e. they act
    // from k F dt on.  In loop terms: the chain runs at phase F-1 on the current state and opens a chunk of up to F
    // ticks whose first RK4 step still runs with the old commands; the new ones are latched after that step.  The
    // tick at t = 0 finds messages nobody has written (zeros); a tick that coincides with the end of a launch belongs
    // to that launch (ExecuteSimulation runs the tasks scheduled at its stop time).  nav_lag = 0: the chain runs at
    // phase 0 on the state of its own time, latched at once.
    bool navlag = false;
    if constexpr (NRW > 0) navlag = a.nav_lag != 0;
    // The FSW output messages: the newest wheel torque command and thruster burst.  The effectors' copies (u, ev.thr_*,
    // thr_t0) are refreshed from them by plain moves (`latch`) - idempotent, so no "new message" flags are carried.
    double un[NRW > 0 ? NRW : 1];
#pragma unroll
    for (int k = 0; k < NRW; ++k) un[k] = u[k];
    unsigned lim2n[BSK_MAX_THR / 2] = {0u, 0u, 0u, 0u};
    int thr_maxn = 0, thr_t0n = 0;
    if constexpr (FULL) {
#pragma unroll
        for (int k = 0; k < BSK_MAX_THR / 2; ++k) lim2n[k] = ev.thr_lim2[k];
        thr_maxn = ev.thr_max;
        thr_t0n = thr_t0;
    }
    auto fsw_tick = [&](const State<NRW>& nav, int t_latch) {
        // mrpControlTask order of the reference (MRP_Feedback before attTrackingError, ...Simulator.py:484-486;
        // bsk_config.fsw_lag): this tick commands the torque the PREVIOUS tick's guidance maps to and
        // leaves its own for the next one.  The pending torque comes from the slab on the launch's first
        // FSW tick (issued here, consumed after the guidance arithmetic) and stays in registers afterwards.
        // (fsw_lag is a kernel argument: read from the cold block it cost the chain a memory round trip of its own)
        const bool lag = a.fsw_lag != 0;
        if (lag && first_fsw) {
#pragma unroll
            for (int k = 0; k < NRW; ++k) up[k] = ldf(st + (int64_t)(TAIL + BSK_T_UPEND + k) * S, bo);
        }
        const FswCfg fc = load_fsw(cold);
        Guid g = guidance<NRW>(fc.sigma_R0N, nav, action);
        sbr = sqrt_nr(dot(g.sigma_BR, g.sigma_BR));
        if (lag) {
#pragma unroll
            for (int k = 0; k < NRW; ++k) un[k] = up[k];
            control<NRW>(fc, g, up);
        } else {
            control<NRW>(fc, g, un);
        }
        fsw_ran = true;
        if constexpr (FULL) {
            if (desat && action == 2) {
                bool fired = false;
                desat_tick<NRW>(cold, nav.Om, first_fsw, fsw_every * c.h, 2.0 / c.h, fsw_every,
                                const_cast<double*>(st) + (int64_t)(TAIL + BSK_T_THR_REM) * S, S, bo, lim2n, fired, thr_cnt);
                if (fired) {   // the burst starts when the thruster set latches the on-time message
                    thr_maxn = 0;
#pragma unroll
                    for (int k = 0; k < BSK_MAX_THR; ++k) thr_maxn = max(thr_maxn, (int)((lim2n[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
                    thr_t0n = t_latch;
                }
            }
        }
        first_fsw = false;
    };
    // the dynamics task's effectors read the FSW output messages: the newest torque / burst act from the current tick on
    auto latch = [&]() {
#pragma unroll
        for (int k = 0; k < NRW; ++k) u[k] = un[k];
        if constexpr (FULL) {
#pragma unroll
            for (int k = 0; k < BSK_MAX_THR / 2; ++k) ev.thr_lim2[k] = lim2n[k];
            ev.thr_max = thr_maxn;
            thr_t0 = thr_t0n;
        }
    };
    // the FSW tasks of t = 0 (nav_lag): nothing has written their inputs yet.  Handled as a chunk of zero ticks at the head
    // of the loop (wave-uniform: the other lanes of the wave wait), so that the FSW chain is instantiated once.
    bool z0 = false;
    if constexpr (NRW > 0) z0 = navlag && tick == 0 && substeps_eff > 0;
    unsigned long long dbg_waitA = 0, dbg_waitB = 0, dbg_chain = 0, dbg_bar = 0;      // (probe builds only: bsk_probes.hpp)
    if constexpr (PAIR) {
        static_assert(!PAIR || (FEAT >= FEAT_POWER && FEAT != FEAT_FULLG && GRAV != BSK_GRAV_SH), "pair form: power / full-scenario levels, point mass or J2");
        static_assert(!TRI || FEAT == FEAT_FULL, "three-wave form: the full-scenario level");
        // ---- pair form: both waves run the SAME control flow on the same counters (every decision below is a function of
        // cnt / substeps / actions, identical in the two waves), so their barrier counts agree by construction; what each
        // wave does between two barriers depends on its role and contains no barrier.
        PairP PL = (PairP)lds_dyn;
        // wave 0: dynamics, wave 1: FSW + environment - swapped in every other group of 2^pair_shift workgroups, so that the
        // two waves a SIMD hosts (they come from different workgroups) are one of each kind
        // (three-wave form: wave 0 rotational half, wave 1 FSW + environment, wave 2 translational half)
        const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const bool isD = TRI ? wave_id != 1 : ((wave_id ^ (int)(blockIdx.x >> a.pair_shift)) & 1) == 0;
        TriXP TX = nullptr;
        if constexpr (TRI) TX = (TriXP) & ((TriP)lds_dyn)->x;
        if (!isD) {
            PL->sun[0][lane] = sg.sun.x; PL->sun[1][lane] = sg.sun.y; PL->sun[2][lane] = sg.sun.z;
            PL->lext[0][lane] = lext.x; PL->lext[1][lane] = lext.y; PL->lext[2][lane] = lext.z;
            if (lane == 0) PL->qcount = 0;
            if constexpr (TRI) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { TX->tv[k][lane] = 0; TX->ts[k][lane] = 0; }   // (LDS keeps the previous launch's tags)
                if (lane == 0) TX->err = 0;
            }
        }
        const double draw = a.power.draw, cap = a.power.cap;
        // the environment wave's share of a chunk: eclipse -> panel of each recorded tick, cooperative drain of the
        // partially eclipsed ones, battery updates in tick order (the arithmetic of power_tick / power_flush)
        auto env_ticks = [&](int mm, int bb) {
            if (mm == 0) return;
            double gk[PAIR_CHUNK], sk[PAIR_CHUNK];
            unsigned bandmask = 0u;
            auto eval = [&](int k) __attribute__((always_inline)) {
                const V3 r = mk(PL->rr[bb][0][k][lane], PL->rr[bb][1][k][lane], PL->rr[bb][2][k][lane]);
                const V3 sig = mk(PL->rs[bb][0][k][lane], PL->rs[bb][1][k][lane], PL->rs[bb][2][k][lane]);
                bool band;
                power_eval<FULL>(a.power, sg, r, sig, kt.c, gk[k], sk[k], band);
                bandmask |= band ? (1u << k) : 0u;
            };
            if (mm == PAIR_CHUNK) {                            // a whole chunk: the evaluations are independent chains in ONE
#pragma unroll                                                 // scheduling region (each alone is a serial chain through rsq / rcp)
                for (int k = 0; k < PAIR_CHUNK; ++k) eval(k);
            } else {
#pragma unroll
                for (int k = 0; k < PAIR_CHUNK; ++k) {
                    gk[k] = 0.0; sk[k] = 1.0;
                    if (k < mm) eval(k);
                }
            }
            if (BSK_UNLIKELY(__builtin_amdgcn_ballot_w64(bandmask != 0u) != 0)) {
#pragma unroll
                for (int k = 0; k < PAIR_CHUNK; ++k) {
                    if (bandmask & (1u << k)) {
                        const int e = __hip_atomic_fetch_add(&PL->qcount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        PL->qown[e] = lane | (k << 8);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int qc = PL->qcount;
                for (int e = lane; e < qc; e += 64) {          // entry e by lane e mod 64, whoever owns it
                    const int own = PL->qown[e], ol = own & 63, k = own >> 8;
                    const V3 r = mk(PL->rr[bb][0][k][ol], PL->rr[bb][1][k][ol], PL->rr[bb][2][k][ol]);
                    const V3 sun = mk(PL->sun[0][ol], PL->sun[1][ol], PL->sun[2][ol]);
                    PL->sfac[k][ol] = percent_shadow(a.power, sun - r, r, dot(r, r));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int k = 0; k < PAIR_CHUNK; ++k) {
                    if (bandmask & (1u << k)) sk[k] = PL->sfac[k][lane];
                }
                if (lane == 0) PL->qcount = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
#pragma unroll
            for (int k = 0; k < PAIR_CHUNK; ++k) {
                if (k < mm) {
                    shadow = sk[k];
                    charge = fmin(fmax(fma(fma(gk[k], sk[k], draw), c.h, charge), 0.0), cap);   // (the arithmetic of power_flush)
                }
            }
        };
        // The commands of the FSW wave (box rows 0..9), read by the dynamics wave STRAIGHT into its effectors' copies
        // (u, ev.thr_*, thr_t0): that is the latch, and no second copy of the messages stays live in its tick loop (which
        // has 256 registers: every one it does not hold is a scratch reload less per tick).
        auto read_cmd = [&]() {
#pragma unroll
            for (int k = 0; k < NRW; ++k) u[k] = PL->box[k][lane];
            if constexpr (FULL) {
#pragma unroll
                for (int k = 0; k < BSK_MAX_THR / 2; ++k) ev.thr_lim2[k] = (unsigned)PL->box[4 + k][lane];
                ev.thr_max = (int)PL->box[8][lane];
                thr_t0 = (int)PL->box[9][lane];
            }
        };
        // One chunk's control decisions: a function of (phase, z0, tick, j) only - both waves evaluate it on identical
        // inputs, so their barrier sequences agree.
        struct Chunk { int m; bool cond, fsw_any, anyz, zlane, needB, needB0; int t_latch; };
        auto next_chunk = [&]() {
            Chunk q;
            q.m = substeps_eff - j;
            q.cond = false; q.fsw_any = false; q.anyz = false; q.zlane = false; q.t_latch = tick;
            if constexpr (NRW > 0) {
                const int trig = navlag ? fsw_every - 1 : 0;
                int dist = trig - phase;
                if (dist <= 0) dist += fsw_every;
                q.anyz = navlag && __builtin_amdgcn_ballot_w64(z0) != 0;
                q.cond = z0 || (!q.anyz && phase == trig);
                q.fsw_any = __builtin_amdgcn_ballot_w64(q.cond) != 0;
                q.zlane = z0;
                q.t_latch = tick + ((navlag && !z0) ? 1 : 0);
                if (q.cond && navlag) dist = fsw_every;   // latched inside the chunk, after its first RK4 step
                if (q.anyz) dist = 0;                      // t = 0 tick: latched without a step in between
                z0 = false;
                q.m = min(q.m, dist);
            }
            q.m = wave_min_uniform(min(q.m, PAIR_CHUNK));
            if constexpr (NRW > 0) {
                phase += q.m;
                if (phase >= fsw_every) phase -= fsw_every;
            }
            j += q.m;
            q.needB0 = q.fsw_any && !navlag;               // same-tick chain: commands before the chunk's first step
            q.needB = q.fsw_any && navlag;                 // reference priorities: commands latched after the first step
            return q;
        };
        int cb = 0;                                        // ring buffer of this chunk
        // (probe builds with BSK_PROBE_TRI_ROLE time every barrier; stamp<false> / since<false> are empty: the product's macro is the barrier)
#define BSK_PAIR_SYNC() do { const probe::Stamp b0_ = probe::stamp<probe::TRI_ROLE>(); __syncthreads(); probe::since<probe::TRI_ROLE>(dbg_bar, b0_); } while (0)
        BSK_PAIR_SYNC();                                   // the environment wave's Sun positions are in LDS
        const probe::Stamp role_t0 = probe::stamp<probe::TRI_ROLE>();
        dbg_bar = 0;
        // role probe: {cycles in the tick loop, of them at barriers, `third`} of the probed role's wave -> the workgroup's debug word
        auto role_word = [&](unsigned long long third) {
            if constexpr (TRI && probe::TRI_ROLE) {
                if (wave_id == probe::TRI_ROLE_WAVE && lane == 0 && a.tail.dbg) a.tail.dbg[blockIdx.x] = probe::pack3(probe::elapsed(role_t0), dbg_bar, third);
            }
        };
        if constexpr (TRI) {
          if (isD) {
            // ------------------------------------------------- rotational / translational wave (one loop per role: a shared
            // loop would keep the union of both halves' registers live)
            auto dyn_part = [&](auto ROLE) __attribute__((always_inline)) {
                constexpr int PART = decltype(ROLE)::value;
                constexpr bool ROT = PART == PART_ROT;
                TriXch<PART> xc{TX, lane, 0, 0, false};
                // exponentialAtmosphere at a position (the arithmetic of the single-wave tick loop), 0 below the skip density
                // (anchored where the single-wave loop anchors: the launch's first tick, every chunk's first tick - whose
                // density the last step of the chunk before it asks for)
                bool atm_anchor = true;
                auto density = [&](V3 r) __attribute__((always_inline)) {
                    double rho = 0.0;
                    if (drag_cfg) {
                        const double r2 = dot(r, r);
                        const double rm = r2 * rsqrt_nr(r2);
                        if (atm_anchor) rho = atm.anchor(kt, rm);
                        else rho = atm.advance(kt, rm);
                    }
                    return rho;
                };
                // the first tick's first-stage values (every later tick's are published by the step before it)
                if constexpr (ROT) xc.template publish<false>(x.s, 0.0);
                else { ev.rho = density(x.r); xc.template publish<true>(x.v, ev.rho); }
                while (j < substeps_eff) {
                    const Chunk q = next_chunk();
                    const int m = q.m;
                    if (q.fsw_any) {                       // each half's share of the navigation / wheel-speed messages
                        if constexpr (ROT) {
                            PL->box[6][lane] = x.s.x; PL->box[7][lane] = x.s.y; PL->box[8][lane] = x.s.z;
                            PL->box[9][lane] = x.w.x; PL->box[10][lane] = x.w.y; PL->box[11][lane] = x.w.z;
#pragma unroll
                            for (int k = 0; k < NRW; ++k) PL->box[12 + k][lane] = x.Om[k];
                        } else {
                            PL->box[0][lane] = x.r.x; PL->box[1][lane] = x.r.y; PL->box[2][lane] = x.r.z;
                            PL->box[3][lane] = x.v.x; PL->box[4][lane] = x.v.y; PL->box[5][lane] = x.v.z;
                        }
                    }
                    BSK_PAIR_SYNC();                       // A
                    if (q.needB0) {
                        BSK_PAIR_SYNC();                   // B (same-tick chain)
                        read_cmd();
                    }
                    if constexpr (!ROT) {
                        if (ev.sun_on) third_body_anchor(ev.s3, mk(PL->sun[0][lane], PL->sun[1][lane], PL->sun[2][lane]), a.extra.mu_sun, x.r);
                    }
                    for (int t = 0; t < m; ++t, ++tick) {
                        V3 v1 = mk(0, 0, 0), Tw = mk(0, 0, 0), pw = mk(0, 0, 0);
                        double tq[NRW > 0 ? NRW : 1], tqj[NRW > 0 ? NRW : 1];
                        if constexpr (ROT) {
                            // the first-stage velocity and the tick's density are on their way while the wheels' terms are formed
                            const TriPend f = xc.template prefetch<true>();
#pragma unroll
                            for (int i = 0; i < NRW; ++i) {
                                const double fs = fmin(fmax(x.Om[i] * 0x1p1000, -c.fc), c.fc);
                                tq[i] = u[i] - fs;
                            }
                            if constexpr (NRW > 0) { pw = mv3<DIAG>(c.W, x.w); wv.head(tq, x.Om, Tw, pw, tqj); }
                            double rho = 0.0;
                            v1 = xc.template finish<true>(f, rho);
                            ev.rho = rho;
                        }
                        ev.drag_on = __builtin_amdgcn_ballot_w64(ev.rho != 0.0) != 0;
                        if (desat) {
                            ev.e2 = 2 * (tick - thr_t0);
                            ev.thr_on = __builtin_amdgcn_ballot_w64(ev.thr_max > 0 && ev.e2 <= ev.thr_max) != 0;
                            if (BSK_UNLIKELY(ev.thr_on)) thr_masks(ev);
                        }
                        const double tt = (double)tick * c.h;
                        double rho_next = 0.0;
                        atm_anchor = t == m - 1;
#define BSK_TRI_STEP(THR, DRAGM) rk4_step_part<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DRAGM, PART>(c, wv, x, lext, tt, ev, xc, v1, tq, Tw, pw, tqj, density, rho_next)
                        if (BSK_LIKELY(!ev.thr_on)) {
                            if (BSK_LIKELY(ev.drag_on)) BSK_TRI_STEP(false, 1);
                            else BSK_TRI_STEP(false, 2);
                        } else BSK_TRI_STEP(true, 0);
#undef BSK_TRI_STEP
                        if constexpr (!ROT) ev.rho = rho_next;
                        if constexpr (ROT) {
                            PL->rs[cb][0][t][lane] = x.s.x; PL->rs[cb][1][t][lane] = x.s.y; PL->rs[cb][2][t][lane] = x.s.z;
                        } else {
                            PL->rr[cb][0][t][lane] = x.r.x; PL->rr[cb][1][t][lane] = x.r.y; PL->rr[cb][2][t][lane] = x.r.z;
                        }
                        if (t == 0 && q.needB) {           // B: the commands of this chunk's FSW tick are there
                            BSK_PAIR_SYNC();
                            read_cmd();
                        }
                    }
                    if (m == 0 && q.needB) {               // the t = 0 chunk has no step
                        BSK_PAIR_SYNC();
                        read_cmd();
                    }
                    cb ^= 1;
                }
                if constexpr (!ROT) {                      // the rotational wave writes the launch's results: it gets (r, v)
                    PL->box[10][lane] = x.r.x; PL->box[11][lane] = x.r.y; PL->box[12][lane] = x.r.z;
                    PL->box[13][lane] = x.v.x; PL->box[14][lane] = x.v.y; PL->box[15][lane] = x.v.z;
                }
                if constexpr (probe::TRI_XCHG != 0) {      // exchange probe: misses (16 bits) | re-reads (16) | cycles / 64 spent re-reading (32), per wave
                    const unsigned long long w = (unsigned long long)(xc.dbg_miss & 0xFFFFu) | ((unsigned long long)(xc.dbg_spin & 0xFFFFu) << 16) | (((xc.dbg_cyc >> 6) & 0xFFFFFFFFull) << 32);
                    if (lane == 0) ((volatile unsigned long long __attribute__((address_space(3)))*)&TX->pad_[0])[ROT ? 0 : 1] = w;
                }
                BSK_PAIR_SYNC();                           // the last chunk's ring is complete
                BSK_PAIR_SYNC();                           // ... and the environment wave has answered
                role_word(xc.dbg_cyc);
            };
            if (wave_id == 0) dyn_part(std::integral_constant<int, PART_ROT>{});
            else { dyn_part(std::integral_constant<int, PART_TRA>{}); return; }
            x.r = mk(PL->box[10][lane], PL->box[11][lane], PL->box[12][lane]);
            x.v = mk(PL->box[13][lane], PL->box[14][lane], PL->box[15][lane]);
            charge = PL->box[0][lane]; shadow = PL->box[1][lane]; sbr = PL->box[2][lane];
#pragma unroll
            for (int k = 0; k < NRW; ++k) up[k] = PL->box[3 + k][lane];
            thr_cnt = (int)PL->box[7][lane];
            fsw_ran = PL->box[8][lane] != 0.0;
            tri_failed = TX->err != 0;     // an exchange timed out (cannot happen): reported through the handle's error word below
          }
        }
        if (isD && !TRI) {
            // ---------------------------------------------------------------- dynamics wave
            while (j < substeps_eff) {
                const Chunk q = next_chunk();
                const int m = q.m;
                if (q.fsw_any) {                           // the navigation / wheel-speed messages of this FSW tick
                    PL->box[0][lane] = x.r.x; PL->box[1][lane] = x.r.y; PL->box[2][lane] = x.r.z;
                    PL->box[3][lane] = x.v.x; PL->box[4][lane] = x.v.y; PL->box[5][lane] = x.v.z;
                    PL->box[6][lane] = x.s.x; PL->box[7][lane] = x.s.y; PL->box[8][lane] = x.s.z;
                    PL->box[9][lane] = x.w.x; PL->box[10][lane] = x.w.y; PL->box[11][lane] = x.w.z;
#pragma unroll
                    for (int k = 0; k < NRW; ++k) PL->box[12 + k][lane] = x.Om[k];
                }
                const probe::Stamp wa0 = probe::stamp<probe::PAIR_WAIT>();
                BSK_PAIR_SYNC();                           // A: messages written; the previous chunk's ring complete
                probe::since<probe::PAIR_WAIT>(dbg_waitA, wa0);
                if (q.needB0) {
                    BSK_PAIR_SYNC();                       // B (same-tick chain): wait for the commands
                    read_cmd();
                }
                if constexpr (FULL) {
                    if (ev.sun_on) third_body_anchor(ev.s3, mk(PL->sun[0][lane], PL->sun[1][lane], PL->sun[2][lane]), a.extra.mu_sun, x.r);
                }
                if constexpr (FULL) {
                    if (drag_cfg) { const double r2 = dot(x.r, x.r); atm.anchor(kt, r2 * rsqrt_nr(r2)); }
                }
                for (int t = 0; t < m; ++t, ++tick) {
                    const V3 lx = mk(PL->lext[0][lane], PL->lext[1][lane], PL->lext[2][lane]);   // (parked in LDS)
                    if constexpr (FULL) {
                        if (drag_cfg) {
                            const double r2 = dot(x.r, x.r);
                            ev.rho = atm.advance(kt, r2 * rsqrt_nr(r2));
                            ev.drag_on = __builtin_amdgcn_ballot_w64(ev.rho != 0.0) != 0;
                        }
                        if (desat) {
                            ev.e2 = 2 * (tick - thr_t0);
                            ev.thr_on = __builtin_amdgcn_ballot_w64(ev.thr_max > 0 && ev.e2 <= ev.thr_max) != 0;
                            if (BSK_UNLIKELY(ev.thr_on)) thr_masks(ev);
                        }
                        if (BSK_LIKELY(!ev.thr_on)) {
                            if (BSK_LIKELY(ev.drag_on)) rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, false, 1>(c, wv, x, u, lx, (double)tick * c.h, ev, accp);
                            else rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, false, 2>(c, wv, x, u, lx, (double)tick * c.h, ev, accp);
                        } else rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, true, 0>(c, wv, x, u, lx, (double)tick * c.h, ev, accp);
                    } else {
                        rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, false>(c, wv, x, u, lx, (double)tick * c.h, ev, accp);
                    }
                    PL->rr[cb][0][t][lane] = x.r.x; PL->rr[cb][1][t][lane] = x.r.y; PL->rr[cb][2][t][lane] = x.r.z;
                    PL->rs[cb][0][t][lane] = x.s.x; PL->rs[cb][1][t][lane] = x.s.y; PL->rs[cb][2][t][lane] = x.s.z;
                    if (t == 0 && q.needB) {               // B: the commands of this chunk's FSW tick are there
                        const probe::Stamp w0 = probe::stamp<probe::PAIR_WAIT>();
                        BSK_PAIR_SYNC();
                        probe::since<probe::PAIR_WAIT>(dbg_waitB, w0);
                        read_cmd();
                    }
                }
                if (m == 0 && q.needB) {                   // the t = 0 chunk has no step
                    BSK_PAIR_SYNC();
                    read_cmd();
                }
                cb ^= 1;
            }
            BSK_PAIR_SYNC();                               // the last chunk's ring is complete
            BSK_PAIR_SYNC();                               // ... and the environment wave has answered
            charge = PL->box[0][lane]; shadow = PL->box[1][lane]; sbr = PL->box[2][lane];
#pragma unroll
            for (int k = 0; k < NRW; ++k) up[k] = PL->box[3 + k][lane];
            thr_cnt = (int)PL->box[7][lane];
            fsw_ran = PL->box[8][lane] != 0.0;
        } else if (!isD) {
            // ---------------------------------------------------------------- FSW + environment wave
            // It has a fifth of the other wave's instructions and sits on its critical path (commands at every FSW tick, a ring
            // buffer per chunk): where it shares a SIMD with a dynamics wave it must not queue behind it.
            __builtin_amdgcn_s_setprio(3);
            int pm = 0, pb = 0;                            // length / buffer of the chunk whose EnvTask ticks are still owed
            while (j < substeps_eff) {
                const Chunk q = next_chunk();
                BSK_PAIR_SYNC();                           // A
                const probe::Stamp c0 = probe::stamp<probe::PAIR_WAIT || probe::TRI_ROLE>();
                if (q.fsw_any) {
                    State<NRW> nav;
                    nav.r = mk(PL->box[0][lane], PL->box[1][lane], PL->box[2][lane]);
                    nav.v = mk(PL->box[3][lane], PL->box[4][lane], PL->box[5][lane]);
                    nav.s = mk(PL->box[6][lane], PL->box[7][lane], PL->box[8][lane]);
                    nav.w = mk(PL->box[9][lane], PL->box[10][lane], PL->box[11][lane]);
#pragma unroll
                    for (int k = 0; k < NRW; ++k) nav.Om[k] = PL->box[12 + k][lane];
                    if (BSK_UNLIKELY(q.anyz)) {
                        if (q.zlane) {
                            nav.r = mk(0, 0, 0); nav.v = mk(0, 0, 0); nav.s = mk(0, 0, 0); nav.w = mk(0, 0, 0);
#pragma unroll
                            for (int k = 0; k < NRW; ++k) nav.Om[k] = 0.0;
                        }
                    }
                    if constexpr (NRW > 0) {
                        if (q.cond) fsw_tick(nav, q.t_latch);
                    }
#pragma unroll
                    for (int k = 0; k < NRW; ++k) PL->box[k][lane] = un[k];
                    if constexpr (FULL) {
#pragma unroll
                        for (int k = 0; k < BSK_MAX_THR / 2; ++k) PL->box[4 + k][lane] = (double)lim2n[k];
                        PL->box[8][lane] = (double)thr_maxn;
                        PL->box[9][lane] = (double)thr_t0n;
                    }
                    probe::since<probe::PAIR_WAIT || probe::TRI_ROLE>(dbg_chain, c0);
                    BSK_PAIR_SYNC();                       // B (either timing): the commands are there
                }
                tick += q.m;
                env_ticks(pm, pb);                         // the PREVIOUS chunk's EnvTask, beside this chunk's integration
                pm = q.m; pb = cb;
                cb ^= 1;
            }
            BSK_PAIR_SYNC();                               // the last chunk's ring is complete
            env_ticks(pm, pb);
            PL->box[0][lane] = charge; PL->box[1][lane] = shadow; PL->box[2][lane] = sbr;
#pragma unroll
            for (int k = 0; k < NRW; ++k) PL->box[3 + k][lane] = up[k];
            PL->box[7][lane] = (double)thr_cnt;
            PL->box[8][lane] = fsw_ran ? 1.0 : 0.0;
            if constexpr (probe::PAIR_WAIT) PL->box[10][lane] = (double)(dbg_chain >> 4);
            if constexpr (probe::PAIR_HWID) PL->box[9][lane] = (double)probe::hw_id();
            BSK_PAIR_SYNC();
            role_word(dbg_chain);
            return;                                        // the dynamics wave writes the launch's results
        }
    } else {
    int np = 0;   // power system: ticks recorded since the last flush (per lane)
    PowerCfg pc_loop_cfg;   // power level (panel constants as scalar operands): the four the tick reads, as scalars of their own
    if constexpr (POWER && !FULL) {
        pc_loop_cfg = a.power;
#pragma unroll
        for (int k = 0; k < 3; ++k) pc_loop_cfg.nB[k] = own_scalar(a.power.nB[k]);
        pc_loop_cfg.kflux = own_scalar(a.power.kflux);
    }
    const probe::Stamp pc_loop = probe::stamp<probe::CHUNK != 0>();
    while (j < substeps_eff) {
        const probe::Stamp pc_head = probe::stamp<probe::CHUNK == 1>();
        int m = substeps_eff - j;
        bool fsw_here = false;                         // this lane's FSW chain ran at the head of this chunk
        if constexpr (NRW > 0) {
            const int trig = navlag ? fsw_every - 1 : 0;
            int dist = trig - phase;                   // ticks to the next FSW tick of this lane
            if (dist <= 0) dist += fsw_every;
            const bool anyz = navlag && __builtin_amdgcn_ballot_w64(z0) != 0;
            if (z0 || (!anyz && phase == trig)) {
                State<NRW> nav = x;
                if (BSK_UNLIKELY(anyz)) {              // wave-uniform: the per-lane selects only exist at t = 0
                    if (z0) {
                        nav.r = mk(0, 0, 0); nav.v = mk(0, 0, 0); nav.s = mk(0, 0, 0); nav.w = mk(0, 0, 0);
#pragma unroll
                        for (int k = 0; k < NRW; ++k) nav.Om[k] = 0.0;
                    }
                }
                const probe::Stamp pc_fsw = probe::stamp<probe::CHUNK == 3>();
                fsw_tick(nav, tick + ((navlag && !z0) ? 1 : 0));
                probe::since<probe::CHUNK == 3>(dbg_chain, pc_fsw);
                fsw_here = true;
                if (!navlag) latch();
                else dist = fsw_every;                 // latched inside the chunk, after its first RK4 step
            }
            if (anyz) dist = 0;                        // t = 0 tick: latched below without a step in between
            z0 = false;
            m = min(m, dist);
            if constexpr (POWER) m = min(m, PEN_CHUNK);
            // The DPP-broadcast harmonics need every lane active inside the RK4 loop, so the trip count is
            // made wave-uniform: envs of one wave that sit at different FSW phases (after a masked reset)
            // advance together to the nearest FSW tick of any of them.
            // (the full-scenario levels take their wave-uniform constants through DPP broadcasts: same requirement;
            // and every level with the power system: the penumbra queue is drained cooperatively - entry e by lane
            // e mod 64, whoever owns it - so a flush must find all 64 lanes in the same loop iteration)
            if constexpr (GRAV == BSK_GRAV_SH || WDPP || POWER) m = wave_min_uniform(m);
            phase += m;
            if (phase >= fsw_every) phase -= fsw_every;
        }
        if constexpr (POWER && NRW == 0) {
            m = min(m, PEN_CHUNK);
            if constexpr (FULL) m = wave_min_uniform(m);
        }
        if constexpr (POWER) {
            // the tick record is flushed (queue drained cooperatively, battery replayed) when some lane's would overflow
            const probe::Stamp pc_flush = probe::stamp<probe::CHUNK == 4>();
            if (__builtin_amdgcn_ballot_w64(np + m > PEN_SLOTS) != 0) {
                power_flush(a.power, L, np, lane, c.h, charge, shadow);
                np = 0;
            }
            probe::since<probe::CHUNK == 4>(dbg_chain, pc_flush);
        }
        j += m;
        if constexpr (FULL) {
            const probe::Stamp pc_anchor = probe::stamp<probe::CHUNK == 5>();
            if (ev.sun_on) third_body_anchor(ev.s3, sg.sun, a.extra.mu_sun, x.r);   // exact at the chunk's first position
            if (drag_cfg_u) atm.anchor(kt, pre.r2 * rsqrt_nr(pre.r2));             // ... and the density's full evaluation
            probe::since<probe::CHUNK == 5>(dbg_chain, pc_anchor);
        }
        // ---- the chunk's ticks.  A tick = head (what decides which instantiation of the step runs: the atmosphere's density,
        // the burst test) + body (RK4 step, EnvTask).  Three instantiations at the full-scenario levels: with drag (nearly
        // always in LEO below ~460 km), without, inside a thruster burst (rare).
        // The run's control flags live in SCALAR registers as integers (uni()): as `bool`s the compiler keeps wave-uniform
        // conditions as lane masks and tests them with a v_cndmask + v_cmp + s_andn2 triple per branch.
        int burst_any = 0;           // some thruster of some lane inside a burst (wave-uniform, settled per chunk: below)
        int drag_now = 0, thr_now = 0;   // this tick: some lane inside the atmosphere / inside a burst
        int t = 0;
        auto tick_head = [&]() __attribute__((always_inline)) {
            if constexpr (FULL) {
                pre.ir = rsqrt_nr(pre.r2);                   // 1 / |r|: the atmosphere below, the first stage's gravity
                if (drag_cfg_u) {   // exponentialAtmosphere, refreshed once per dyn tick
                    ev.rho = atm.advance(kt, pre.r2 * pre.ir);
                    drag_now = uni_any(ev.rho != 0.0);   // any lane of the wave inside the atmosphere
                }
                // some thruster of some lane still inside its burst
                if (BSK_UNLIKELY(burst_any)) thr_now = uni_any(ev.thr_max > 0 && 2 * (tick - thr_t0) <= ev.thr_max);
            }
        };
        auto tick_body = [&](auto THR_, auto DM_) __attribute__((always_inline)) {
            constexpr bool THR = decltype(THR_)::value;
            constexpr int DM = decltype(DM_)::value;
            double q2 = 0.0;         // |sigma|^2 after the step (full-scenario levels: the EnvTask's rotation starts from it)
            if constexpr (FULL && THR) {      // (only here: the burst's masks and rows must not be carried around the other runs' loops)
                ev.e2 = 2 * (tick - thr_t0);
                ev.drag_on = drag_now != 0;       // (this instantiation tests the drag switch per stage)
                thr_masks(ev);
            }
            if constexpr (FULL) rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, THR, DM>(c, wv, x, u, lext, (double)tick * c.h, ev, accp, &pre, &q2);
            else rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, false>(c, wv, x, u, lext, (double)tick * c.h, ev, accp);
            if constexpr (POWER) power_tick<FULL>(FULL ? a.power : pc_loop_cfg, sg, x.r, x.s, L, np + t, lane, kt.c, FULL ? &pre : nullptr, FULL ? &q2 : nullptr);
            ++t;
            ++tick;
        };
        using B0 = std::integral_constant<bool, false>;
        using B1 = std::integral_constant<bool, true>;
        using I0_ = std::integral_constant<int, 0>;
        using I1_ = std::integral_constant<int, 1>;
        using I2_ = std::integral_constant<int, 2>;
        // Is some thruster of some lane inside a burst?  Bursts begin only where the dynamics task latches an on-time message
        // (latch(), after a chunk's first step) and last one control period at most, so the question is settled per chunk -
        // here for its first step, again after the latch - and the ticks of a wave without a burst (nearly all) skip the
        // per-tick test (eleven instructions) altogether.
        auto burst_pending = [&]() __attribute__((always_inline)) {
            int any = 0;
            if constexpr (FULL) {
                if (desat) any = uni_any(ev.thr_max > 0 && 2 * (tick - thr_t0) <= ev.thr_max);
                if (!any) thr_now = 0;
            }
            return any;
        };
        burst_any = burst_pending();
        probe::since<probe::CHUNK == 1>(dbg_chain, pc_head);
        const probe::Stamp pc_first = probe::stamp<probe::CHUNK == 2>();
        // The chunk's first tick runs alone when an FSW tick opened the chunk under the reference's task priorities: after it
        // the dynamics task's effectors latch the new commands (idempotent moves, a no-op for lanes without a new message).
        // No other tick latches anything.
        if constexpr (NRW > 0) {
            bool lat = navlag && fsw_here;
            if constexpr (GRAV == BSK_GRAV_SH || WDPP || POWER) lat = __builtin_amdgcn_ballot_w64(lat) != 0;   // wave-uniform trip counts
            if (lat && m > 0) {
                tick_head();
                if constexpr (FULL) {
                    if (BSK_LIKELY(!thr_now)) {
                        if (BSK_LIKELY(drag_now)) tick_body(B0{}, I1_{});
                        else tick_body(B0{}, I2_{});
                    } else tick_body(B1{}, I0_{});
                } else tick_body(B0{}, I0_{});
                latch();
                burst_any = burst_pending();
            }
        }
        probe::since<probe::CHUNK == 2>(dbg_chain, pc_first);
        // The rest of the chunk as RUNS of ticks of one instantiation, two ticks per trip of the run's loop.  A loop whose body
        // is one tick cannot produce the step's results in the registers its header expects (the old state is read until the
        // last stage: ~30 register moves per tick); a loop that picks the instantiation per tick merges three alternative
        // results after every step (nine more).  A run's loop contains one instantiation only, its second copy ping-pongs
        // between two register sets, and it ends when the next tick's head asks for another kind (or the chunk is through).
        auto run = [&](auto THR_, auto DM_) __attribute__((always_inline)) {      // -> the next tick's head has been evaluated
            constexpr bool THR = decltype(THR_)::value;
            constexpr int DM = decltype(DM_)::value;
            auto same = [&]() __attribute__((always_inline)) {
                if constexpr (!FULL) return true;
                else if constexpr (THR) return false;                              // bursts: one tick at a time
                else return !thr_now && drag_now == (DM == 1 ? 1 : 0);
            };
#pragma nounroll
            for (;;) {
                tick_body(THR_, DM_);
                if (t >= m) return false;
                tick_head();
                if (BSK_UNLIKELY(!same())) return true;
                tick_body(THR_, DM_);
                if (t >= m) return false;
                tick_head();
                if (BSK_UNLIKELY(!same())) return true;
            }
        };
        if constexpr (POWER) {
            bool headed = false;
#pragma nounroll
            while (t < m) {
                if (!headed) tick_head();
                if constexpr (FULL) {
                    if (BSK_LIKELY(!thr_now)) {
                        if (BSK_LIKELY(drag_now)) headed = run(B0{}, I1_{});
                        else headed = run(B0{}, I2_{});
                    } else headed = run(B1{}, I0_{});
                } else headed = run(B0{}, I0_{});
            }
        } else {
            // the bare levels have one instantiation and no head: pairs, then the odd tick (measured: the run machinery costs
            // them 5 VALU instructions per tick, 1 - 2 %)
#pragma nounroll
            while (t + 1 < m) {
                tick_body(B0{}, I0_{});
                tick_body(B0{}, I0_{});
            }
            if (t < m) tick_body(B0{}, I0_{});
        }
        np += m;
        if constexpr (NRW > 0) latch();        // (the t = 0 chunk has no step: its commands are latched here)
    }
    if constexpr (POWER) {
        if (__builtin_amdgcn_ballot_w64(np > 0) != 0) power_flush(a.power, L, np, lane, c.h, charge, shadow);
    }
    if constexpr (probe::CHUNK != 0) dbg_waitA = probe::elapsed(pc_loop);
    }   // !PAIR

    // Re-read the post-loop arguments from the kernarg segment through an opaque pointer: the
    // compiler cannot hoist these scalar loads above the loop, so they cost it no SGPRs.
    typedef const TailArgs __attribute__((address_space(4))) * TailPtr;
    TailPtr tp = (TailPtr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                           offsetof(StepArgs<NRW __COMMA__ DIAG>, tail));
    asm volatile("" : "+s"(tp));
    // ... and in ONE batch: fetched field by field where they are used, the ~11 scalar loads of the epilogue were
    // eleven dependent round trips on the critical path of a K = 1 launch (a wave alone on its SIMD waits each out)
    TailArgs ta;
    __builtin_memcpy(&ta, (const TailArgs*)tp, sizeof(TailArgs));
    const int64_t S2 = uniform64(ta.stride);
    const int64_t SO = uniform64(ta.ostride);      // observation rows: unpadded (the slab's rows are: bsk_capi.hip, bsk_create)
    const int n2 = ta.n;
    const bool valid2 = gid < n2;
    if constexpr (TRI) {
        // the three-wave exchange gave up on a value: the launch's results are not to be trusted.  NaN observations AND
        // the handle's error word, which every synchronising entry point checks (bsk_get_obs / bsk_sync -> BSK_EHIP)
        if (BSK_UNLIKELY(tri_failed)) {
            sbr = __builtin_nan(""); charge = sbr;
            if (lane == 0) __hip_atomic_store(ta.err, BSK_DEVERR_TRI_EXCHANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    // observation: [|sigma_BR|, |omega_BN|, |Omega|/limit, charge/3600/power_max, shadow]
    // obs[0] is the logged att_guidance message: with nav_lag the one the last FSW tick wrote (held in `sbr`),
    // otherwise the tracking error of the end-of-step state under the step's mode
    double o0 = sbr;
    if (!(NRW > 0 && ta.nav_lag != 0)) {
        double sR0N[3] = {ta.obs_cfg.sigma_R0N[0], ta.obs_cfg.sigma_R0N[1], ta.obs_cfg.sigma_R0N[2]};
        const Guid g = guidance<NRW>(sR0N, x, action);
        o0 = sqrt_nr(dot(g.sigma_BR, g.sigma_BR));
    }
    const double o1 = sqrt_nr(dot(x.w, x.w));
    double om2 = 0.0;
#pragma unroll
    for (int k = 0; k < NRW; ++k) om2 = fma(x.Om[k], x.Om[k], om2);
    const double o2 = sqrt_nr(om2) * ta.obs_cfg.inv_wheel_limit;
    const bool static_o3 = FEAT < FEAT_POWER && ta.static_charge != 0;     // obs[3] already sits in the buffers (wave-uniform)
    double o3 = charge * ta.obs_cfg.charge_scale;
    const double o4 = shadow;

    // reward and termination
    int why = 0;
    double rew = (action == 0) ? ta.obs_cfg.reward_mult * rcp_nr(fma(o0, o0, 1.0)) : 0.0;
    if (steps0 >= ta.obs_cfg.max_length) why |= BSK_DONE_LENGTH;
    if (o2 > 1.0) { why |= BSK_DONE_WHEELS; rew -= ta.obs_cfg.failure_penalty; }
    if (!static_o3 && o3 == 0.0) { why |= BSK_DONE_BATTERY; rew -= ta.obs_cfg.failure_penalty; }
    if (dot(x.r, x.r) < ta.obs_cfg.r_min2) why |= BSK_DONE_ORBIT;

    if constexpr (SPLIT == 5) {
        if (threadIdx.x & 64) return;    // the second wave of each pair only helped with the harmonics
    }
    // wavefront reductions (every lane of the wave participates; tail lanes contribute nothing)
    const unsigned long long dmask = __ballot(valid2 && why != 0);
    if constexpr (probe::ANY) {
        // probe builds (bsk_probes.hpp): one 64-bit word per wave in the handle's debug buffer - never in the done mask
        unsigned long long w = 0ull;
        if constexpr (TRI && probe::TRI_XCHG != 0)      // the rotational wave's exchange word (1) or the translational wave's (2)
            w = ((volatile unsigned long long __attribute__((address_space(3)))*)&((TriP)lds_dyn)->x.pad_[0])[probe::TRI_XCHG == 2 ? 1 : 0];
        if constexpr (PAIR && probe::PAIR_WAIT)         // cycles / 16 the dynamics wave waited at B (bits 0-20), at A (21-41), the other wave's chain time (42-62)
            w = ((dbg_waitB >> 4) & 0x1FFFFFull) | (((dbg_waitA >> 4) & 0x1FFFFFull) << 21) | (((unsigned long long)((PairP)lds_dyn)->box[10][0] & 0x1FFFFFull) << 42);
        if constexpr (PAIR && probe::PAIR_TIME)         // residency: this wave's cycles / 1024 | its hardware id << 32
            w = ((probe::elapsed(t_kernel) >> 10) & 0xFFFFFFFFull) | (probe::hw_id() << 32);
        if constexpr (PAIR && probe::PAIR_HWID)         // placement: the dynamics wave's hardware id | the environment wave's << 32
            w = probe::hw_id() | ((unsigned long long)((PairP)lds_dyn)->box[9][0] << 32);
        if constexpr (!PAIR && probe::CHUNK != 0)       // single-wave form: cycles / 16 of the probed part of every chunk | of the whole tick loop << 32
            w = ((dbg_chain >> 4) & 0xFFFFFFFFull) | (((dbg_waitA >> 4) & 0xFFFFFFFFull) << 32);
        if ((threadIdx.x & 63) == 0 && ta.dbg && !(TRI && probe::TRI_ROLE)) ta.dbg[gid >> 6] = w;      // (the role probe's wave wrote the word itself)
    }
    if ((threadIdx.x & 63) == 0) {
        ta.done_mask[gid >> 6] = dmask;
    }

    // Tail lanes shadow env n-1 and computed bit-identical results from identical inputs, so their
    // stores (same address, same value) need no mask.
    gptr<double> so = uniform_ptr(ta.st);
    gptr<double> ob = uniform_ptr(ta.obs);
#define FLD(f) (so + (int64_t)(f) * S2)
    const int n_pool = ta.n_pool;
    if (n_pool > 0 && why != 0) {
      if (valid2) {   // tail lanes shadow env n-1: its own lane performs the reset, they must not repeat it
        // Device-side auto-reset (rare, divergent): reload this env from the staged IC pool, keep the
        // finished episode's observation as terminal observation, report the new episode's first one.
        gptr<double> tob = uniform_ptr(ta.term_obs);
        if (static_o3) o3 = *(gptr<double>)((gptr<char>)(ob + 3 * SO) + bo);      // (rare path: the finished episode's constant obs[3])
        stf(tob + 0 * SO, bo, o0); stf(tob + 1 * SO, bo, o1); stf(tob + 2 * SO, bo, o2); stf(tob + 3 * SO, bo, o3);
        stf(tob + 4 * SO, bo, o4);
        const int ep = ta.episodes[i];
        ta.episodes[i] = ep + 1;
        const unsigned slot = (((unsigned)i + ta.env_base) * 2654435761u + (unsigned)ep * 40503u + 12345u) % (unsigned)n_pool;
        const double* __restrict__ pool = ta.pool;
        const int nf = ta.n_fields;
        for (int f = 0; f < nf; ++f) stf(FLD(f), bo, pool[(int64_t)f * n_pool + slot]);
        const V3 ps = mk(pool[(int64_t)(BSK_F_SIGMA + 0) * n_pool + slot], pool[(int64_t)(BSK_F_SIGMA + 1) * n_pool + slot],
                         pool[(int64_t)(BSK_F_SIGMA + 2) * n_pool + slot]);
        const V3 pw = mk(pool[(int64_t)(BSK_F_OMEGA + 0) * n_pool + slot], pool[(int64_t)(BSK_F_OMEGA + 1) * n_pool + slot],
                         pool[(int64_t)(BSK_F_OMEGA + 2) * n_pool + slot]);
        double pom2 = 0.0;
#pragma unroll
        for (int k = 0; k < NRW; ++k) {
            const double v = pool[(int64_t)(BSK_NF_BASE + k) * n_pool + slot];
            pom2 = fma(v, v, pom2);
        }
        const double n0 = sqrt_nr(dot(ps, ps)), n1 = sqrt_nr(dot(pw, pw)), n2 = sqrt_nr(pom2) * ta.obs_cfg.inv_wheel_limit;
        const double n3 = pool[(int64_t)(TAIL + BSK_T_CHARGE) * n_pool + slot] * ta.obs_cfg.charge_scale;
        stf(ob + 0 * SO, bo, n0); stf(ob + 1 * SO, bo, n1); stf(ob + 2 * SO, bo, n2); stf(ob + 3 * SO, bo, n3);
        stf(ob + 4 * SO, bo, 1.0);
        if (ta.obs_rm) {
            double* __restrict__ rm = ta.obs_rm + (int64_t)i * 5;
            rm[0] = n0; rm[1] = n1; rm[2] = n2; rm[3] = n3; rm[4] = 1.0;
        }
        *(gptr<unsigned long long>)((gptr<char>)uniform_ptr(ta.cnt) + bo) = 0ull;
      }
    } else {
        stf(FLD(BSK_F_R + 0), bo, x.r.x); stf(FLD(BSK_F_R + 1), bo, x.r.y); stf(FLD(BSK_F_R + 2), bo, x.r.z);
        stf(FLD(BSK_F_V + 0), bo, x.v.x); stf(FLD(BSK_F_V + 1), bo, x.v.y); stf(FLD(BSK_F_V + 2), bo, x.v.z);
        stf(FLD(BSK_F_SIGMA + 0), bo, x.s.x); stf(FLD(BSK_F_SIGMA + 1), bo, x.s.y); stf(FLD(BSK_F_SIGMA + 2), bo, x.s.z);
        stf(FLD(BSK_F_OMEGA + 0), bo, x.w.x); stf(FLD(BSK_F_OMEGA + 1), bo, x.w.y); stf(FLD(BSK_F_OMEGA + 2), bo, x.w.z);
#pragma unroll
        for (int k = 0; k < NRW; ++k) stf(FLD(BSK_NF_BASE + k), bo, x.Om[k]);
        if constexpr (POWER) stf(FLD(TAIL + BSK_T_CHARGE), bo, charge);
        if constexpr (FULL) {
            if (desat) {
#pragma unroll
                for (int k = 0; k < BSK_MAX_THR; ++k) stf(FLD(TAIL + BSK_T_THR_LIM + k), bo, (double)thr_limit(ev, k));
                stf(FLD(TAIL + BSK_T_THR_T0), bo, (double)thr_t0);
                stf(FLD(TAIL + BSK_T_THR_CNT), bo, (double)thr_cnt);
            }
        }
        if constexpr (NRW > 0) {
            if (fsw_ran) {
#pragma unroll
                for (int k = 0; k < NRW; ++k) stf(FLD(TAIL + BSK_T_UCMD + k), bo, u[k]);
                if (ta.fsw_lag) {
#pragma unroll
                    for (int k = 0; k < NRW; ++k) stf(FLD(TAIL + BSK_T_UPEND + k), bo, up[k]);
                }
                stf(FLD(TAIL + BSK_T_SBR), bo, sbr);
            }
        }
        // int2 {steps | phase << 20, ticks} written as one 8-byte word
        // the step count saturates at 2^20 - 1 so that it can never spill into the phase bits
        const unsigned long long packed = (unsigned long long)(unsigned)(min(steps0 + 1, 0xFFFFF) | (phase << 20)) |
                                          ((unsigned long long)(unsigned)(cnt.y + ta.substeps) << 32);
        *(gptr<unsigned long long>)((gptr<char>)uniform_ptr(ta.cnt) + bo) = packed;
        stf(ob + 0 * SO, bo, o0); stf(ob + 1 * SO, bo, o1); stf(ob + 2 * SO, bo, o2);
        if (!static_o3) stf(ob + 3 * SO, bo, o3);
        stf(ob + 4 * SO, bo, o4);
        if (ta.obs_rm) {       // optional row-major copy (N, 5): what a torch policy reshapes without a copy kernel
            double* __restrict__ rm = ta.obs_rm + (int64_t)i * 5;
            rm[0] = o0; rm[1] = o1; rm[2] = o2; rm[4] = o4;
            if (!static_o3) rm[3] = o3;
        }
    }
#undef FLD
    stf(uniform_ptr(ta.reward), bo, rew);
    ta.reason[i] = (unsigned char)why;
    if (ta.ep_return) {
        // Monitor-style episode statistics on the device (reference envs/leoPowerAttitudeEnvironment.py:130-135:
        // 'r' = the episode's rewards including this step's, 'l' = env steps taken BEFORE this one)
        ep_ret += rew;
        if (why != 0) {
            stf(uniform_ptr(ta.term_return), bo, ep_ret);
            ta.term_len[i] = steps0;
            if (n_pool > 0) ep_ret = 0.0;          // the device-side reset above started a new episode
        }
        stf(uniform_ptr(ta.ep_return), bo, ep_ret);
        ta.done[i] = why != 0 ? 1 : 0;
    }
    if (BSK_UNLIKELY(ta.wave_sum != nullptr)) {
        // bsk_set_step_stats: the first level of the batch reduction (stats_kernel, bsk_aux.hip) done here, behind every store of
        // the launch - the same butterfly over the same 64 rewards, so the scalars do not depend on who formed the wave sums
        const double ws = wave_sum(valid2 ? rew : 0.0);
        if ((threadIdx.x & 63) == 0) ta.wave_sum[gid >> 6] = ws;
    }
}

template <int GRAV, int NRW, bool DIAG, int FEAT, int SPLIT>
static hipError_t launch_t(const StepParams& p, const StepBuffers& b, int block, hipStream_t s, hipEvent_t ev0,
                           hipEvent_t ev1) {
    StepArgs<NRW, DIAG> a;
    fill_hot<GRAV, NRW, DIAG>(p, a.hot);
    a.cold = b.cold; a.st = b.st; a.cnt = b.cnt; a.act = b.act;
    a.stride = b.stride; a.n = b.n; a.substeps = b.substeps;
    a.nav_lag = p.nav_lag; a.fsw_lag = p.fsw_lag;
    a.pair_shift = p.pair_shift; a.act_shift = b.act_shift; a.ep_return = b.ep_return;
    a.static_charge = b.static_charge; a.pad2_ = 0;
    a.power = p.pc;
    a.extra = p.ex;
    a.tail.obs_cfg = p.obs; a.tail.st = b.st; a.tail.cnt = b.cnt; a.tail.obs = b.obs; a.tail.reward = b.reward;
    a.tail.done_mask = b.done_mask; a.tail.reason = b.reason;
    a.tail.stride = b.stride; a.tail.ostride = b.ostride; a.tail.n = b.n; a.tail.substeps = b.substeps;
    a.tail.pool = b.pool; a.tail.term_obs = b.term_obs; a.tail.episodes = b.episodes;
    a.tail.n_pool = b.n_pool; a.tail.n_fields = b.n_fields;
    a.tail.fsw_lag = p.fsw_lag; a.tail.nav_lag = p.nav_lag;
    a.tail.env_base = b.env_base; a.tail.static_charge = b.static_charge;
    a.tail.ep_return = b.ep_return; a.tail.term_return = b.term_return; a.tail.term_len = b.term_len; a.tail.done = b.done;
    a.tail.obs_rm = b.obs_rm; a.tail.err = b.err; a.tail.dbg = b.dbg; a.tail.wave_sum = b.wave_sum;
    if (SPLIT == 5) block = 256;
    if (SPLIT == 2) block = 128;      // pair form: dynamics wave + FSW / environment wave of the same 64 spacecraft
    if (SPLIT == 3) block = 192;      // three-wave form: rotational, FSW / environment and translational wave
    const int grid = SPLIT == 5 ? (b.n + 127) / 128 : ((SPLIT == 2 || SPLIT == 3) ? (b.n + 63) / 64 : (b.n + block - 1) / block);
    // the power system keeps a per-wave tick record and penumbra queue in dynamic LDS (bsk_device.hpp: PowerLds)
    const size_t lds = SPLIT == 3 ? sizeof(TriLds) : SPLIT == 2 ? sizeof(PairLds)
                     : FEAT >= FEAT_POWER ? sizeof(PowerLds) * (size_t)(block / 64)
                                          : (FEAT == FEAT_LDSS ? sizeof(AccLds) * (size_t)(block / 64) : 0);
    // hipExtLaunchKernelGGL stamps ev0/ev1 from the dispatch packet itself (no marker packets), so
    // their difference is the kernel's own duration, as rocprofv3 --kernel-trace reports it.
    if (lds > 48 * 1024) {   // the two-wave harmonics form with the power system: 4 waves x 29 KB of dynamic LDS
        // the attribute belongs to (kernel, DEVICE): one bit per device and instantiation, set once the call has
        // succeeded there (handles on different devices are stepped from different threads: atomic, and a lost race
        // only repeats an idempotent call)
        static std::atomic<unsigned long long> raised{0ull};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute((const void*)&step_kernel<GRAV, NRW, DIAG, FEAT, SPLIT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    hipExtLaunchKernelGGL((step_kernel<GRAV, NRW, DIAG, FEAT, SPLIT>), dim3(grid), dim3(block), lds, s, ev0, ev1, 0, a);
    return hipGetLastError();
}

// ---- which instantiations this translation unit holds ----------------------------------------------------------------------
// A cell = one (gravity model, hub kind, feature level) with its three wheel sets (and every form of it: harmonics forms 4 / 5,
// pair / three-wave forms).  The library is built from EIGHT compilations of this file (-DBSK_TU=0..7, make -j: about one minute
// instead of six), each holding the cells below - balanced by measured compile time, the harmonics' scenario levels being the
// expensive ones (~45 s a cell against 2 - 15 s) - and bsk_dispatch.hip trying them in turn.  Without BSK_TU the file holds
// everything it is asked for in ONE unit: the fast builds of the variant / probe libraries and of tests/test_dpp_build.py.
#define BSK_CELL(X, G, D, P) X(G, 0, D, P) X(G, 3, D, P) X(G, 4, D, P)
#define BSK_UNIT0(X) BSK_CELL(X, BSK_GRAV_SH, true, 3)
#define BSK_UNIT1(X) BSK_CELL(X, BSK_GRAV_SH, true, 2)
#define BSK_UNIT2(X) BSK_CELL(X, BSK_GRAV_SH, false, 3)
#define BSK_UNIT3(X) BSK_CELL(X, BSK_GRAV_SH, false, 2)
#define BSK_UNIT4(X) BSK_CELL(X, BSK_GRAV_SH, true, 0) BSK_CELL(X, BSK_GRAV_SH, true, 1) BSK_CELL(X, BSK_GRAV_SH, false, 0)
#define BSK_UNIT5(X) BSK_CELL(X, BSK_GRAV_SH, false, 1) BSK_CELL(X, BSK_GRAV_PM, true, 2) BSK_CELL(X, BSK_GRAV_PM_J2, true, 2)
#define BSK_UNIT6(X) BSK_CELL(X, BSK_GRAV_PM_J2, true, 0) BSK_CELL(X, BSK_GRAV_PM_J2, true, 1) BSK_CELL(X, BSK_GRAV_PM_J2, true, 3)   \
                   BSK_CELL(X, BSK_GRAV_PM_J2, false, 0) BSK_CELL(X, BSK_GRAV_PM_J2, false, 1) BSK_CELL(X, BSK_GRAV_PM_J2, false, 2) \
                   BSK_CELL(X, BSK_GRAV_PM_J2, false, 3) BSK_CELL(X, BSK_GRAV_PM_J2, true, -1) BSK_CELL(X, BSK_GRAV_PM_J2, false, -1)
#define BSK_UNIT7(X) BSK_CELL(X, BSK_GRAV_PM, true, 0) BSK_CELL(X, BSK_GRAV_PM, true, 1) BSK_CELL(X, BSK_GRAV_PM, true, 3)            \
                   BSK_CELL(X, BSK_GRAV_PM, false, 0) BSK_CELL(X, BSK_GRAV_PM, false, 1) BSK_CELL(X, BSK_GRAV_PM, false, 2)          \
                   BSK_CELL(X, BSK_GRAV_PM, false, 3) BSK_CELL(X, BSK_GRAV_PM, true, -1) BSK_CELL(X, BSK_GRAV_PM, false, -1)
#if defined(BSK_TU)
#define BSK_TU_CAT2(a, b) a##b
#define BSK_TU_CAT(a, b) BSK_TU_CAT2(a, b)
#define BSK_VARIANTS(X) BSK_TU_CAT(BSK_UNIT, BSK_TU)(X)
#define BSK_LAUNCH_UNIT BSK_TU_CAT(launch_step_tu, BSK_TU)
#define BSK_PTR_UNIT BSK_TU_CAT(step_kernel_ptr_tu, BSK_TU)
#else
#define BSK_LAUNCH_UNIT launch_step_unit
#define BSK_PTR_UNIT step_kernel_ptr_unit
#if defined(BSK_ONLY)                                // whatever the command line lists: -D'BSK_ONLY(X)=X(BSK_GRAV_SH,4,true,2)'
#define BSK_VARIANTS(X) BSK_ONLY(X)
#elif defined(BSK_FAST_BUILD) && BSK_FAST_BUILD == 3 // only the kernel the drop-in env runs, in its three forms (tests/test_dpp_build.py)
#define BSK_VARIANTS(X) X(BSK_GRAV_PM_J2, 3, true, 2)
#elif defined(BSK_FAST_BUILD) && BSK_FAST_BUILD == 2 // only the harmonics kernels of the config-5 bench
#define BSK_VARIANTS(X) X(BSK_GRAV_SH, 4, true, 0) X(BSK_GRAV_SH, 4, true, 1) X(BSK_GRAV_SH, 4, true, 2) X(BSK_GRAV_SH, 4, true, 3)
#elif defined(BSK_FAST_BUILD)                        // ISA inspection / A-B builds: the J2 + 4-wheel (bench) and J2 + 3-wheel (env) kernels
#define BSK_FAST_LEVELS(X, R) X(BSK_GRAV_PM_J2, R, true, 0) X(BSK_GRAV_PM_J2, R, true, 1) X(BSK_GRAV_PM_J2, R, true, 2) X(BSK_GRAV_PM_J2, R, true, 3) X(BSK_GRAV_PM_J2, R, true, -1)
#define BSK_VARIANTS(X) BSK_FAST_LEVELS(X, 4) BSK_FAST_LEVELS(X, 3)
#else                                                // everything in one unit
#define BSK_VARIANTS(X) BSK_UNIT0(X) BSK_UNIT1(X) BSK_UNIT2(X) BSK_UNIT3(X) BSK_UNIT4(X) BSK_UNIT5(X) BSK_UNIT6(X) BSK_UNIT7(X)
#endif
#endif

// pair form (SPLIT == 2): built for the power / full-scenario levels of the point-mass and J2 kernels with a diagonal hub
template <int G, int R, bool D, int P>
static hipError_t launch_pair(const StepParams& p, const StepBuffers& b, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if constexpr (D && G != BSK_GRAV_SH && (P == FEAT_POWER || P == FEAT_FULL)) return launch_t<G, R, D, P, 2>(p, b, 128, s, ev0, ev1);
    else return hipErrorInvalidValue;
}
template <int G, int R, bool D, int P>
static const void* pair_ptr() {
    if constexpr (D && G != BSK_GRAV_SH && (P == FEAT_POWER || P == FEAT_FULL)) return (const void*)&step_kernel<G, R, D, P, 2>;
    else return nullptr;
}
// three-wave form (SPLIT == 3): the full-scenario level of the same kernels
template <int G, int R, bool D, int P>
static hipError_t launch_tri(const StepParams& p, const StepBuffers& b, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if constexpr (D && G != BSK_GRAV_SH && P == FEAT_FULL) return launch_t<G, R, D, P, 3>(p, b, 192, s, ev0, ev1);
    else return hipErrorInvalidValue;
}
template <int G, int R, bool D, int P>
static const void* tri_ptr() {
    if constexpr (D && G != BSK_GRAV_SH && P == FEAT_FULL) return (const void*)&step_kernel<G, R, D, P, 3>;
    else return nullptr;
}

// this unit's share of the dispatch: *handled says whether the configuration is one of its instantiations
hipError_t BSK_LAUNCH_UNIT(int grav, int nrw, bool diag, int feat, const StepParams& p, const StepBuffers& b, int block,
                           hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool* handled) {
    *handled = true;
#define CASE(G, R, D, P) \
    if (grav == G && nrw == R && diag == D && feat == P) {                                                 \
        if (p.tri) return launch_tri<G, R, D, P>(p, b, s, ev0, ev1);                                        \
        if (p.pair) return launch_pair<G, R, D, P>(p, b, s, ev0, ev1);                                      \
        if (G == BSK_GRAV_SH && p.sh_form == 4) return launch_t<G, R, D, P, (G == BSK_GRAV_SH ? 4 : 1)>(p, b, block, s, ev0, ev1); \
        if (G == BSK_GRAV_SH && p.sh_form == 5) return launch_t<G, R, D, P, (G == BSK_GRAV_SH ? 5 : 1)>(p, b, block, s, ev0, ev1); \
        return launch_t<G, R, D, P, 1>(p, b, block, s, ev0, ev1);                                           \
    }
    BSK_VARIANTS(CASE)
#undef CASE
    *handled = false;
    return hipErrorInvalidValue;
}

const void* BSK_PTR_UNIT(int grav, int nrw, bool diag, int feat, int sh_form, bool pair, bool tri, bool* handled) {
    *handled = true;
#define CASE(G, R, D, P) \
    if (grav == G && nrw == R && diag == D && feat == P) {                                                                     \
        if (tri) return tri_ptr<G, R, D, P>();                                                                                 \
        if (pair) return pair_ptr<G, R, D, P>();                                                                               \
        if (G == BSK_GRAV_SH && sh_form == 4) return (const void*)&step_kernel<G, R, D, P, (G == BSK_GRAV_SH ? 4 : 1)>;        \
        if (G == BSK_GRAV_SH && sh_form == 5) return (const void*)&step_kernel<G, R, D, P, (G == BSK_GRAV_SH ? 5 : 1)>;        \
        return (const void*)&step_kernel<G, R, D, P, 1>;                                                                       \
    }
    BSK_VARIANTS(CASE)
#undef CASE
    *handled = false;
    return nullptr;
}

#ifndef BSK_TU
// one-unit builds (variant / probe libraries): the dispatcher is this unit itself
hipError_t launch_step(int grav, int nrw, bool diag, int feat, const StepParams& p, const StepBuffers& b, int block,
                       hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    bool handled = false;
    return launch_step_unit(grav, nrw, diag, feat, p, b, block, s, ev0, ev1, &handled);
}
const void* step_kernel_ptr(int grav, int nrw, bool diag, int feat, int sh_form, bool pair, bool tri) {
    bool handled = false;
    return step_kernel_ptr_unit(grav, nrw, diag, feat, sh_form, pair, tri, &handled);
}
bool pair_available(int grav, bool diag, int feat) { return diag && grav != BSK_GRAV_SH && (feat == FEAT_POWER || feat == FEAT_FULL); }
bool tri_available(int grav, bool diag, int feat) { return diag && grav != BSK_GRAV_SH && feat == FEAT_FULL; }
#endif

}  // namespace bsk
