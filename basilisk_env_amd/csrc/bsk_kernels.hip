// bsk_kernels.hip — gfx950 kernels of the batched propagator.
//
// step_kernel<GRAV, NRW>: one spacecraft per lane, 64-lane wavefronts.
//   HBM layout: structure-of-arrays fp64, field f of env i at st[f*stride + i]; a wave reads 512
//   contiguous bytes per field (global_load_dwordx2 per lane, fully coalesced), state lives in
//   VGPRs for all `substeps` RK4 steps, and is written back once.  Reward / done are reduced per
//   wavefront: __ballot gives the 64-bit done mask (one store per wave), a shuffle tree gives
//   the wave's reward sum (one store per wave) — no atomics, bitwise reproducible.
//
// Replaces run_sim + reward/done logic for N spacecraft:
//   reference basilisk_env/simulators/leoPowerAttitudeSimulator.py:535-644
//   reference basilisk_env/envs/leoPowerAttitudeEnvironment.py:98-127,161-170
#include "bsk_device.hpp"
#include "bsk_launch.hpp"

namespace bsk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int GRAV, int NRW>
__global__ __launch_bounds__(256) void step_kernel(const StepArgs a) {
    const DevCfg& c = a.c;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = gid < a.n;
    const int i = valid ? gid : a.n - 1;  // tail lanes shadow the last env; their stores are masked
    const int64_t S = a.stride;
    double* __restrict__ st = a.st;

    State<NRW> x;
    x.r = mk(st[(BSK_F_R + 0) * S + i], st[(BSK_F_R + 1) * S + i], st[(BSK_F_R + 2) * S + i]);
    x.v = mk(st[(BSK_F_V + 0) * S + i], st[(BSK_F_V + 1) * S + i], st[(BSK_F_V + 2) * S + i]);
    x.s = mk(st[(BSK_F_SIGMA + 0) * S + i], st[(BSK_F_SIGMA + 1) * S + i], st[(BSK_F_SIGMA + 2) * S + i]);
    x.w = mk(st[(BSK_F_OMEGA + 0) * S + i], st[(BSK_F_OMEGA + 1) * S + i], st[(BSK_F_OMEGA + 2) * S + i]);
#pragma unroll
    for (int k = 0; k < NRW; ++k) x.Om[k] = st[(BSK_NF_BASE + k) * S + i];
    constexpr int TAIL = BSK_NF_BASE + NRW;
    const V3 lext = mk(st[(TAIL + BSK_T_LEXT + 0) * S + i], st[(TAIL + BSK_T_LEXT + 1) * S + i],
                       st[(TAIL + BSK_T_LEXT + 2) * S + i]);
    const double charge = st[(TAIL + BSK_T_CHARGE) * S + i];
    const int2 cnt = a.cnt[i];
    const int action = a.act[i];

    int phase = cnt.y % c.fsw_every;
    double u[NRW > 0 ? NRW : 1];
    if constexpr (NRW > 0) {
        // the held motor torque only matters when this launch starts between two FSW ticks
        if (phase != 0) {
#pragma unroll
            for (int k = 0; k < NRW; ++k) u[k] = st[(TAIL + BSK_T_UCMD + k) * S + i];
        } else {
#pragma unroll
            for (int k = 0; k < NRW; ++k) u[k] = 0.0;
        }
    }
    bool fsw_ran = false;

    for (int j = 0; j < a.substeps; ++j) {
        if constexpr (NRW > 0) {
            if (phase == 0) {
                Guid g = guidance<NRW>(c, x, action);
                control<NRW>(c, g, u);
                fsw_ran = true;
            }
            phase = (phase + 1 == c.fsw_every) ? 0 : phase + 1;
        }
        rk4_step<GRAV, NRW>(c, x, u, lext);
    }

    // observation: [|sigma_BR|, |omega_BN|, |Omega|/limit, charge/3600/power_max, shadow]
    const Guid g = guidance<NRW>(c, x, action);
    const double o0 = sqrt(dot(g.sigma_BR, g.sigma_BR));
    const double o1 = sqrt(dot(x.w, x.w));
    double om2 = 0.0;
#pragma unroll
    for (int k = 0; k < NRW; ++k) om2 = fma(x.Om[k], x.Om[k], om2);
    const double o2 = sqrt(om2) * c.inv_wheel_limit;
    const double o3 = charge * c.charge_scale;
    const double o4 = 1.0;

    // reward and termination
    int why = 0;
    double rew = (action == 0) ? c.reward_mult / fma(o0, o0, 1.0) : 0.0;
    if (cnt.x >= c.max_length) why |= BSK_DONE_LENGTH;
    if (o2 > 1.0) { why |= BSK_DONE_WHEELS; rew -= c.failure_penalty; }
    if (o3 == 0.0) { why |= BSK_DONE_BATTERY; rew -= c.failure_penalty; }
    if (dot(x.r, x.r) < c.r_min2) why |= BSK_DONE_ORBIT;

    // wavefront reductions (every lane of the wave participates; tail lanes contribute nothing)
    const unsigned long long dmask = __ballot(valid && why != 0);
    const double rsum = wave_sum(valid ? rew : 0.0);
    if ((threadIdx.x & 63) == 0) {
        a.done_mask[gid >> 6] = dmask;
        a.wave_reward[gid >> 6] = rsum;
    }

    if (valid) {
        st[(BSK_F_R + 0) * S + i] = x.r.x; st[(BSK_F_R + 1) * S + i] = x.r.y; st[(BSK_F_R + 2) * S + i] = x.r.z;
        st[(BSK_F_V + 0) * S + i] = x.v.x; st[(BSK_F_V + 1) * S + i] = x.v.y; st[(BSK_F_V + 2) * S + i] = x.v.z;
        st[(BSK_F_SIGMA + 0) * S + i] = x.s.x; st[(BSK_F_SIGMA + 1) * S + i] = x.s.y; st[(BSK_F_SIGMA + 2) * S + i] = x.s.z;
        st[(BSK_F_OMEGA + 0) * S + i] = x.w.x; st[(BSK_F_OMEGA + 1) * S + i] = x.w.y; st[(BSK_F_OMEGA + 2) * S + i] = x.w.z;
#pragma unroll
        for (int k = 0; k < NRW; ++k) st[(BSK_NF_BASE + k) * S + i] = x.Om[k];
        if constexpr (NRW > 0) {
            if (fsw_ran) {
#pragma unroll
                for (int k = 0; k < NRW; ++k) st[(TAIL + BSK_T_UCMD + k) * S + i] = u[k];
            }
        }
        a.cnt[i] = make_int2(cnt.x + 1, cnt.y + a.substeps);
        a.obs[0 * S + i] = o0; a.obs[1 * S + i] = o1; a.obs[2 * S + i] = o2; a.obs[3 * S + i] = o3; a.obs[4 * S + i] = o4;
        a.reward[i] = rew;
        a.reason[i] = (unsigned char)why;
    }
}

// Deterministic batch scalars from the per-wave partials: one 256-thread workgroup, fixed order.
__global__ __launch_bounds__(256) void stats_kernel(const double* __restrict__ wave_reward,
                                                    const unsigned long long* __restrict__ done_mask, int n_waves,
                                                    double* out_sum, long long* out_done) {
    __shared__ double sr[256];
    __shared__ long long sd[256];
    double r = 0.0;
    long long d = 0;
    for (int w = threadIdx.x; w < n_waves; w += 256) {
        r += wave_reward[w];
        d += __popcll(done_mask[w]);
    }
    sr[threadIdx.x] = r;
    sd[threadIdx.x] = d;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sr[threadIdx.x] += sr[threadIdx.x + off];
            sd[threadIdx.x] += sd[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out_sum = sr[0]; *out_done = sd[0]; }
}

// Scatter a compact IC block [nf][m] into the state slab at env indices idx[0..m) and zero their
// counters (bsk_reset with a mask; reference reset / reset_init,
// basilisk_env/envs/leoPowerAttitudeEnvironment.py:172-216).
__global__ void scatter_reset_kernel(double* __restrict__ st, int64_t stride, int nf, const double* __restrict__ ic,
                                     const int* __restrict__ idx, int m, int2* __restrict__ cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int e = idx[t];
    for (int f = 0; f < nf; ++f) st[f * stride + e] = ic[(int64_t)f * m + t];
    cnt[e] = make_int2(0, 0);
}

template <int GRAV, int NRW>
static hipError_t launch_t(const StepArgs& a, int block, hipStream_t s) {
    const int grid = (a.n + block - 1) / block;
    hipLaunchKernelGGL((step_kernel<GRAV, NRW>), dim3(grid), dim3(block), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_step(int grav, int nrw, const StepArgs& a, int block, hipStream_t s) {
#define CASE(G, R) \
    if (grav == G && nrw == R) return launch_t<G, R>(a, block, s);
    CASE(BSK_GRAV_PM, 0) CASE(BSK_GRAV_PM, 3) CASE(BSK_GRAV_PM, 4)
    CASE(BSK_GRAV_PM_J2, 0) CASE(BSK_GRAV_PM_J2, 3) CASE(BSK_GRAV_PM_J2, 4)
#undef CASE
    return hipErrorInvalidValue;
}

const void* step_kernel_ptr(int grav, int nrw) {
#define CASE(G, R) \
    if (grav == G && nrw == R) return (const void*)&step_kernel<G, R>;
    CASE(BSK_GRAV_PM, 0) CASE(BSK_GRAV_PM, 3) CASE(BSK_GRAV_PM, 4)
    CASE(BSK_GRAV_PM_J2, 0) CASE(BSK_GRAV_PM_J2, 3) CASE(BSK_GRAV_PM_J2, 4)
#undef CASE
    return nullptr;
}

hipError_t launch_stats(const double* wave_reward, const unsigned long long* done_mask, int n_waves, double* out_sum,
                        long long* out_done, hipStream_t s) {
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(256), 0, s, wave_reward, done_mask, n_waves, out_sum, out_done);
    return hipGetLastError();
}

hipError_t launch_scatter_reset(double* st, int64_t stride, int nf, const double* ic, const int* idx, int m, int2* cnt,
                                hipStream_t s) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_reset_kernel, dim3((m + 255) / 256), dim3(256), 0, s, st, stride, nf, ic, idx, m, cnt);
    return hipGetLastError();
}

}  // namespace bsk
