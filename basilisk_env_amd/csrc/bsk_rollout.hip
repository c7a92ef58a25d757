// bsk_rollout.hip — open-loop rollouts: T env steps in ONE launch (bsk_step_n).
//
// The reference's own mains step whole episodes with a constant action
//     basilisk_env/envs/leoPowerAttitudeEnvironment.py:218-231          (two episodes of action 0)
//     basilisk_env/simulators/leoPowerAttitudeSimulator.py:657-694      (360 steps of action 0)
// and an open-loop evaluation of a fixed action sequence needs no observation on the host between its steps either.  One launch
// per env step pays the launch's latency chain per step - at K = 1, 4.5 of the headline launch's 6.2 us are what a kernel takes that only streams the same bytes and constants
// plus the state's round trip through memory.  rollout_kernel keeps a spacecraft's state in registers ACROSS env steps: per step
// it reads 4 bytes (the action; nothing for a constant one) and writes 49 (five observations, reward, done reason), the state
// slab is read once and written once per launch.
//
// What one env step does is what step_kernel does at the bare level (mode switch -> FSW chain when due, the reference's task
// order and priorities -> RK4 sub-steps -> observation, reward, done; device-side restart from the staged pool) with the same
// device functions (bsk_device.hpp) in the same order per value: T steps of this kernel leave every buffer of the handle - slab,
// counters, observation / reward / reason / done mask, terminal observations, episode counts and statistics - bit for bit as
// T launches of step_kernel do, and the history rows are what bsk_get_obs would have returned after each of them
// (tests/test_gpu_rollout.py).  Built for the bare propagator (point mass / J2, every wheel set, diagonal and general hub).
#include "bsk_device.hpp"
#include "bsk_launch.hpp"
#include "bsk_rollout.hpp"

#include <cstddef>

namespace bsk {

template <int NRW, bool DIAG>
struct RolloutArgs {
    HotCfg<NRW, DIAG> hot;
    const ColdCfg* cold;
    TailArgs tail;                 // buffers and observation constants, as the step kernel's epilogue takes them
    const int* actions;            // [T][n] device, or NULL: `const_action` at every step
    double* obs_hist;              // [T][5][n]   (any of the three may be NULL)
    double* reward_hist;           // [T][n]
    unsigned char* reason_hist;    // [T][n]
    int n_steps, const_action;
};

// ACT: the launch reads an action per env step and spacecraft (false: ONE action for the whole rollout - no load inside the step loop at
// all).  The actions are fetched a block of ACT_BLOCK steps ahead, as plain loads that stay in flight over the whole block, and packed
// into one register (four bits each) when the block is through: a load per step, consumed by the next step, made the compiler drain the
// memory pipeline at every step's end (s_waitcnt vmcnt(0) at the back edge - the history's seven stores with it: SQ_WAIT_ANY 0.25 of
// the wave's cycles, VALU-active 0.62).
constexpr int ACT_BLOCK = 8;
template <int GRAV, int NRW, bool DIAG, bool ACT>
__global__ __launch_bounds__(256, 2) void rollout_kernel(const RolloutArgs<NRW, DIAG> a) {
    constexpr int FEAT = FEAT_BARE, SPLIT = 1;
    const HotCfg<NRW, DIAG>& c = a.hot;
    const ColdCfg* __restrict__ cold = a.cold;
    const TailArgs& ta = a.tail;
    const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int n = ta.n;
    const bool valid = gid < n;
    const int i = valid ? gid : n - 1;    // tail lanes shadow the last env in registers (identical inputs, identical results); every global store is predicated on `valid`
    const int64_t S = ta.stride;
    const int64_t SO = ta.ostride;
    gptr<double> so = uniform_ptr(ta.st);
    const uint32_t bo = (uint32_t)i * 8u;
    constexpr int TAIL = BSK_NF_BASE + NRW;
#define FLD(f) (so + (int64_t)(f) * S)
    auto ld = [&](int f) __attribute__((always_inline)) { return *(gptr<double>)((gptr<char>)FLD(f) + bo); };

    // ---- the slab, once per launch
    State<NRW> x;
    x.r = mk(ld(BSK_F_R + 0), ld(BSK_F_R + 1), ld(BSK_F_R + 2));
    x.v = mk(ld(BSK_F_V + 0), ld(BSK_F_V + 1), ld(BSK_F_V + 2));
    x.s = mk(ld(BSK_F_SIGMA + 0), ld(BSK_F_SIGMA + 1), ld(BSK_F_SIGMA + 2));
    x.w = mk(ld(BSK_F_OMEGA + 0), ld(BSK_F_OMEGA + 1), ld(BSK_F_OMEGA + 2));
#pragma unroll
    for (int k = 0; k < NRW; ++k) x.Om[k] = ld(BSK_NF_BASE + k);
    V3 lext = mk(ld(TAIL + BSK_T_LEXT + 0), ld(TAIL + BSK_T_LEXT + 1), ld(TAIL + BSK_T_LEXT + 2));
    double charge = ld(TAIL + BSK_T_CHARGE);     // (bare level: constant over an episode)
    double sbr = ld(TAIL + BSK_T_SBR);           // |sigma_BR| of the att_guidance message the last FSW tick wrote
    double u[NRW > 0 ? NRW : 1], up[NRW > 0 ? NRW : 1], un[NRW > 0 ? NRW : 1];
#pragma unroll
    for (int k = 0; k < (NRW > 0 ? NRW : 1); ++k) { u[k] = 0.0; up[k] = 0.0; un[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < NRW; ++k) {              // held torque and the torque the next FSW tick will command: in registers from here on
        u[k] = ld(TAIL + BSK_T_UCMD + k);
        up[k] = ld(TAIL + BSK_T_UPEND + k);
        un[k] = u[k];
    }
    const int2 cnt = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(ta.cnt) + bo);
    int steps0 = cnt.x & 0xFFFFF, phase = cnt.x >> 20, tick = cnt.y;
    const int n_pool = ta.n_pool;
    int ep = (n_pool > 0) ? ta.episodes[i] : 0;  // finished episodes of this env (tracked in a register: shadow lanes never re-read it)
    double ep_ret = 0.0;
    if (ta.ep_return) ep_ret = *(gptr<double>)((gptr<char>)uniform_ptr(ta.ep_return) + bo);

    WheelV<NRW> wv;
    wv.load(c);
    Env ev;                                      // (bare level: never read)
    const int fsw_every = c.fsw_every;
    const int substeps = ta.substeps;
    const bool navlag = NRW > 0 && ta.nav_lag != 0;
    const bool lag = ta.fsw_lag != 0;
    const int hist_n = n;
    double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, o4 = 1.0, rew = 0.0;
    int why = 0;
    bool was_reset = false;                      // the LAST step restarted this env: the slab already holds the new episode

    unsigned act_cur = 0;                       // this block's actions, four bits each
    int act_raw[ACT_BLOCK];                     // the NEXT block's, as loaded
#pragma unroll
    for (int k = 0; k < ACT_BLOCK; ++k) act_raw[k] = 0;
    auto fetch_block = [&](int es0) __attribute__((always_inline)) {
        if constexpr (ACT) {
#pragma unroll
            for (int k = 0; k < ACT_BLOCK; ++k)
                if (es0 + k < a.n_steps) act_raw[k] = a.actions[(int64_t)(es0 + k) * hist_n + i];
        }
    };
    auto pack_block = [&]() __attribute__((always_inline)) {
        unsigned w = 0;
#pragma unroll
        // (four bits each; anything outside 0..14 becomes 15, which no mode test matches - what step_kernel makes of the raw integer)
        for (int k = 0; k < ACT_BLOCK; ++k) w |= ((unsigned)act_raw[k] > 14u ? 15u : (unsigned)act_raw[k]) << (4 * k);
        return w;
    };
    if constexpr (ACT) { fetch_block(0); act_cur = pack_block(); }
    for (int es0 = 0; es0 < a.n_steps; es0 += ACT_BLOCK) {
      if constexpr (ACT) fetch_block(es0 + ACT_BLOCK);
#pragma nounroll
      for (int kk = 0; kk < ACT_BLOCK; ++kk) {
        const int es = es0 + kk;
        if (es >= a.n_steps) break;
        const int action = ACT ? (int)((act_cur >> (4 * kk)) & 15u) : a.const_action;
        // ---- one env step: the tick loop of step_kernel at the bare level
        bool z0 = false;
        if constexpr (NRW > 0) z0 = navlag && tick == 0 && substeps > 0;
        int j = 0;
        auto fsw_tick = [&](const State<NRW>& nav) {
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
        };
        auto latch = [&]() {
#pragma unroll
            for (int k = 0; k < NRW; ++k) u[k] = un[k];
        };
        while (j < substeps) {
            int m = substeps - j;
            bool fsw_here = false;
            if constexpr (NRW > 0) {
                const int trig = navlag ? fsw_every - 1 : 0;
                int dist = trig - phase;
                if (dist <= 0) dist += fsw_every;
                const bool anyz = navlag && __builtin_amdgcn_ballot_w64(z0) != 0;
                if (z0 || (!anyz && phase == trig)) {
                    State<NRW> nav = x;
                    if (BSK_UNLIKELY(anyz)) {
                        if (z0) {
                            nav.r = mk(0, 0, 0); nav.v = mk(0, 0, 0); nav.s = mk(0, 0, 0); nav.w = mk(0, 0, 0);
#pragma unroll
                            for (int k = 0; k < NRW; ++k) nav.Om[k] = 0.0;
                        }
                    }
                    fsw_tick(nav);
                    fsw_here = true;
                    if (!navlag) latch();
                    else dist = fsw_every;
                }
                if (anyz) dist = 0;
                z0 = false;
                m = min(m, dist);
                phase += m;
                if (phase >= fsw_every) phase -= fsw_every;
            }
            j += m;
            int t = 0;
            auto tick_body = [&]() __attribute__((always_inline)) {
                rk4_step<GRAV, NRW, DIAG, FEAT, SPLIT, false>(c, wv, x, u, lext, (double)tick * c.h, ev, nullptr);
                ++t;
                ++tick;
            };
            if constexpr (NRW > 0) {
                if (navlag && fsw_here && m > 0) {
                    tick_body();
                    latch();
                }
            }
#pragma nounroll
            while (t + 1 < m) {
                tick_body();
                tick_body();
            }
            if (t < m) tick_body();
            if constexpr (NRW > 0) latch();
        }

        // ---- observation, reward, termination (step_kernel's epilogue; reference envs/leoPowerAttitudeEnvironment.py:98-127, 161-170)
        o0 = sbr;
        if (!(NRW > 0 && ta.nav_lag != 0)) {
            double sR0N[3] = {ta.obs_cfg.sigma_R0N[0], ta.obs_cfg.sigma_R0N[1], ta.obs_cfg.sigma_R0N[2]};
            const Guid g = guidance<NRW>(sR0N, x, action);
            o0 = sqrt_nr(dot(g.sigma_BR, g.sigma_BR));
        }
        o1 = sqrt_nr(dot(x.w, x.w));
        double om2 = 0.0;
#pragma unroll
        for (int k = 0; k < NRW; ++k) om2 = fma(x.Om[k], x.Om[k], om2);
        o2 = sqrt_nr(om2) * ta.obs_cfg.inv_wheel_limit;
        o3 = charge * ta.obs_cfg.charge_scale;
        o4 = 1.0;
        why = 0;
        rew = (action == 0) ? ta.obs_cfg.reward_mult * rcp_nr(fma(o0, o0, 1.0)) : 0.0;
        if (steps0 >= ta.obs_cfg.max_length) why |= BSK_DONE_LENGTH;
        if (o2 > 1.0) { why |= BSK_DONE_WHEELS; rew -= ta.obs_cfg.failure_penalty; }
        if (o3 == 0.0) { why |= BSK_DONE_BATTERY; rew -= ta.obs_cfg.failure_penalty; }
        if (dot(x.r, x.r) < ta.obs_cfg.r_min2) why |= BSK_DONE_ORBIT;

        // episode statistics in the Monitor convention (envs/leoPowerAttitudeEnvironment.py:130-135)
        if (ta.ep_return) {
            ep_ret += rew;
            if (why != 0) {
                if (valid) {
                    stf(uniform_ptr(ta.term_return), bo, ep_ret);
                    ta.term_len[i] = steps0;
                }
                if (n_pool > 0) ep_ret = 0.0;
            }
        }
        was_reset = false;
        if (n_pool > 0 && why != 0) {
            // device-side restart (rare, divergent): the finished episode's observation is kept as terminal observation, the env
            // continues from pool slot ((env_base + env) 2654435761 + episode 40503 + 12345) mod 2^32 mod n_pool - in the slab (every
            // field, as step_kernel writes it) AND in this lane's registers
            // (shadow lanes of the tail follow env n-1 in registers only: every global store of the restart is its own lane's)
            gptr<double> tob = uniform_ptr(ta.term_obs);
            if (valid) { stf(tob + 0 * SO, bo, o0); stf(tob + 1 * SO, bo, o1); stf(tob + 2 * SO, bo, o2); stf(tob + 3 * SO, bo, o3); stf(tob + 4 * SO, bo, o4); }
            const unsigned slot = (((unsigned)i + ta.env_base) * 2654435761u + (unsigned)ep * 40503u + 12345u) % (unsigned)n_pool;
            ep += 1;
            if (valid) ta.episodes[i] = ep;
            const double* __restrict__ pool = ta.pool;
            const int nf = ta.n_fields;
            if (valid)
                for (int f = 0; f < nf; ++f) stf(FLD(f), bo, pool[(int64_t)f * n_pool + slot]);
            auto pl = [&](int f) { return pool[(int64_t)f * n_pool + slot]; };
            x.r = mk(pl(BSK_F_R + 0), pl(BSK_F_R + 1), pl(BSK_F_R + 2));
            x.v = mk(pl(BSK_F_V + 0), pl(BSK_F_V + 1), pl(BSK_F_V + 2));
            x.s = mk(pl(BSK_F_SIGMA + 0), pl(BSK_F_SIGMA + 1), pl(BSK_F_SIGMA + 2));
            x.w = mk(pl(BSK_F_OMEGA + 0), pl(BSK_F_OMEGA + 1), pl(BSK_F_OMEGA + 2));
            double pom2 = 0.0;
#pragma unroll
            for (int k = 0; k < NRW; ++k) {
                x.Om[k] = pl(BSK_NF_BASE + k);
                pom2 = fma(x.Om[k], x.Om[k], pom2);
                u[k] = pl(TAIL + BSK_T_UCMD + k);
                up[k] = pl(TAIL + BSK_T_UPEND + k);
                un[k] = u[k];
            }
            lext = mk(pl(TAIL + BSK_T_LEXT + 0), pl(TAIL + BSK_T_LEXT + 1), pl(TAIL + BSK_T_LEXT + 2));
            charge = pl(TAIL + BSK_T_CHARGE);
            sbr = pl(TAIL + BSK_T_SBR);
            // what the step reports as observation: the NEW episode's first one (the vec env's convention)
            o0 = sqrt_nr(dot(x.s, x.s)); o1 = sqrt_nr(dot(x.w, x.w)); o2 = sqrt_nr(pom2) * ta.obs_cfg.inv_wheel_limit;
            o3 = charge * ta.obs_cfg.charge_scale; o4 = 1.0;
            steps0 = 0; phase = 0; tick = 0;
            was_reset = true;
        } else {
            steps0 = min(steps0 + 1, 0xFFFFF);       // (saturates: it can never spill into the phase bits)
        }
        // ---- this step's row of the history: 49 bytes per spacecraft
        const int64_t row = (int64_t)es * hist_n + i;
        if (valid && a.obs_hist) {
            double* __restrict__ oh = a.obs_hist + (int64_t)es * 5 * hist_n + i;
            oh[0] = o0; oh[(int64_t)hist_n] = o1; oh[2 * (int64_t)hist_n] = o2; oh[3 * (int64_t)hist_n] = o3; oh[4 * (int64_t)hist_n] = o4;
        }
        if (valid && a.reward_hist) a.reward_hist[row] = rew;
        if (valid && a.reason_hist) a.reason_hist[row] = (unsigned char)why;
      }
      if constexpr (ACT) act_cur = pack_block();
    }

    // ---- the launch's results, where step_kernel leaves them: the last step's outputs, the state, the counters
    const unsigned long long dmask = __ballot(valid && why != 0);
    if ((threadIdx.x & 63) == 0) ta.done_mask[gid >> 6] = dmask;
    if (!valid) return;      // (after the ballot: tail lanes shadowed env n-1 in registers; its own lane stores)
    gptr<double> ob = uniform_ptr(ta.obs);
    stf(ob + 0 * SO, bo, o0); stf(ob + 1 * SO, bo, o1); stf(ob + 2 * SO, bo, o2); stf(ob + 3 * SO, bo, o3); stf(ob + 4 * SO, bo, o4);
    if (ta.obs_rm) {
        double* __restrict__ rm = ta.obs_rm + (int64_t)i * 5;
        rm[0] = o0; rm[1] = o1; rm[2] = o2; rm[3] = o3; rm[4] = o4;
    }
    stf(uniform_ptr(ta.reward), bo, rew);
    ta.reason[i] = (unsigned char)why;
    if (ta.ep_return) {
        stf(uniform_ptr(ta.ep_return), bo, ep_ret);
        ta.done[i] = why != 0 ? 1 : 0;
    }
    if (!was_reset) {
        stf(FLD(BSK_F_R + 0), bo, x.r.x); stf(FLD(BSK_F_R + 1), bo, x.r.y); stf(FLD(BSK_F_R + 2), bo, x.r.z);
        stf(FLD(BSK_F_V + 0), bo, x.v.x); stf(FLD(BSK_F_V + 1), bo, x.v.y); stf(FLD(BSK_F_V + 2), bo, x.v.z);
        stf(FLD(BSK_F_SIGMA + 0), bo, x.s.x); stf(FLD(BSK_F_SIGMA + 1), bo, x.s.y); stf(FLD(BSK_F_SIGMA + 2), bo, x.s.z);
        stf(FLD(BSK_F_OMEGA + 0), bo, x.w.x); stf(FLD(BSK_F_OMEGA + 1), bo, x.w.y); stf(FLD(BSK_F_OMEGA + 2), bo, x.w.z);
#pragma unroll
        for (int k = 0; k < NRW; ++k) {
            stf(FLD(BSK_NF_BASE + k), bo, x.Om[k]);
            stf(FLD(TAIL + BSK_T_UCMD + k), bo, u[k]);        // (unchanged values where no FSW tick ran: the same bits go back)
            stf(FLD(TAIL + BSK_T_UPEND + k), bo, up[k]);
        }
        if constexpr (NRW > 0) stf(FLD(TAIL + BSK_T_SBR), bo, sbr);
    }
    const unsigned long long packed = (unsigned long long)(unsigned)(steps0 | (phase << 20)) | ((unsigned long long)(unsigned)tick << 32);
    *(gptr<unsigned long long>)((gptr<char>)uniform_ptr(ta.cnt) + bo) = packed;
#undef FLD
}

template <int GRAV, int NRW, bool DIAG, bool ACT>
static hipError_t launch_r(const StepParams& p, const StepBuffers& b, const RolloutBuffers& r, int block, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    RolloutArgs<NRW, DIAG> a;
    fill_hot<GRAV, NRW, DIAG>(p, a.hot);
    a.cold = b.cold;
    a.tail.obs_cfg = p.obs; a.tail.st = b.st; a.tail.cnt = b.cnt; a.tail.obs = b.obs; a.tail.reward = b.reward;
    a.tail.done_mask = b.done_mask; a.tail.reason = b.reason;
    a.tail.stride = b.stride; a.tail.ostride = b.ostride; a.tail.n = b.n; a.tail.substeps = b.substeps;
    a.tail.pool = b.pool; a.tail.term_obs = b.term_obs; a.tail.episodes = b.episodes;
    a.tail.n_pool = b.n_pool; a.tail.n_fields = b.n_fields;
    a.tail.fsw_lag = p.fsw_lag; a.tail.nav_lag = p.nav_lag;
    a.tail.env_base = b.env_base; a.tail.static_charge = 0;
    a.tail.ep_return = b.ep_return; a.tail.term_return = b.term_return; a.tail.term_len = b.term_len; a.tail.done = b.done;
    a.tail.obs_rm = b.obs_rm; a.tail.err = b.err; a.tail.dbg = b.dbg; a.tail.wave_sum = nullptr;
    a.actions = r.actions; a.obs_hist = r.obs_hist; a.reward_hist = r.reward_hist; a.reason_hist = r.reason_hist;
    a.n_steps = r.n_steps; a.const_action = r.const_action;
    const int grid = (b.n + block - 1) / block;
    hipExtLaunchKernelGGL((rollout_kernel<GRAV, NRW, DIAG, ACT>), dim3(grid), dim3(block), 0, s, ev0, ev1, 0, a);
    return hipGetLastError();
}

bool rollout_available(int grav, int feat) { return (grav == BSK_GRAV_PM || grav == BSK_GRAV_PM_J2) && feat == FEAT_BARE; }

hipError_t launch_rollout(int grav, int nrw, bool diag, const StepParams& p, const StepBuffers& b, const RolloutBuffers& r, int block,
                          hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#define CASE(G, R, D) if (grav == G && nrw == R && diag == D) return r.actions ? launch_r<G, R, D, true>(p, b, r, block, s, ev0, ev1) : launch_r<G, R, D, false>(p, b, r, block, s, ev0, ev1);
    CASE(BSK_GRAV_PM, 0, true) CASE(BSK_GRAV_PM, 3, true) CASE(BSK_GRAV_PM, 4, true)
    CASE(BSK_GRAV_PM_J2, 0, true) CASE(BSK_GRAV_PM_J2, 3, true) CASE(BSK_GRAV_PM_J2, 4, true)
    CASE(BSK_GRAV_PM, 0, false) CASE(BSK_GRAV_PM, 3, false) CASE(BSK_GRAV_PM, 4, false)
    CASE(BSK_GRAV_PM_J2, 0, false) CASE(BSK_GRAV_PM_J2, 3, false) CASE(BSK_GRAV_PM_J2, 4, false)
#undef CASE
    return hipErrorInvalidValue;
}

const void* rollout_kernel_ptr(int grav, int nrw, bool diag, bool act) {
#define CASE(G, R, D) if (grav == G && nrw == R && diag == D) return act ? (const void*)&rollout_kernel<G, R, D, true> : (const void*)&rollout_kernel<G, R, D, false>;
    CASE(BSK_GRAV_PM, 0, true) CASE(BSK_GRAV_PM, 3, true) CASE(BSK_GRAV_PM, 4, true)
    CASE(BSK_GRAV_PM_J2, 0, true) CASE(BSK_GRAV_PM_J2, 3, true) CASE(BSK_GRAV_PM_J2, 4, true)
    CASE(BSK_GRAV_PM, 0, false) CASE(BSK_GRAV_PM, 3, false) CASE(BSK_GRAV_PM, 4, false)
    CASE(BSK_GRAV_PM_J2, 0, false) CASE(BSK_GRAV_PM_J2, 3, false) CASE(BSK_GRAV_PM_J2, 4, false)
#undef CASE
    return nullptr;
}

}  // namespace bsk
