// bsk_aux.hip — the small kernels around the step kernel (a translation unit of their own: they compile in seconds, the
// step kernel's 36+ instantiations in minutes):
//   stats_kernel            deterministic batch scalars of the last step (sum of rewards, finished envs): SURVEY.md section 8 row a7
//   scatter_reset_kernel    masked reset from host initial conditions (reference reset / reset_init,
//                           basilisk_env/envs/leoPowerAttitudeEnvironment.py:172-216)
//   sample_pool_kernel      on-device IC sampler, Philox4x32-10 (row f4; distributions of
//                           basilisk_env/simulators/leoPowerAttitudeSimulator.py:119-193, leo_orbit.py:25-40, sc_attitudes.py:3-13)
//   reset_from_pool_kernel / init_outputs_kernel    device-side (re)start from the staged pool, first observations
#include "bsk_device.hpp"
#include "bsk_aux.hpp"

#include <algorithm>

namespace bsk {

// (wave_sum(): bsk_device.hpp - the one definition the step kernel's optional in-launch sums share)

// Deterministic batch scalars of the last step: sum of its rewards and number of finished envs (SURVEY.md section 8 row a7:
// "batch sum-reward / sum-done via wave reductions"; reward semantics: reference envs/leoPowerAttitudeEnvironment.py:161-170).
// The step kernel's epilogue carries no reward reduction (a six-stage butterfly through the LDS crossbar on the critical
// path of every launch, for a number asked for once per rollout); these two kernels form the sums from the reward buffer and
// the per-wave done ballots when somebody asks.  Two levels, the order fixed by the batch alone:
//   stats_kernel (many workgroups; hardware wave k of workgroup b takes the step kernel's waves w = 4 b + k, + 4 gridDim, ...):
//     wave w = rewards [64 w, 64 w + 64) summed in the butterfly's order -> wave_sum[w]; the waves' done ballots are popcounted
//     into one integer per workgroup -> done_part[b] (no atomics: when a whole batch finishes together 2 048 workgroups adding
//     to one word took 23 us of a 28 us launch);
//   stats_join_kernel (one workgroup, launched behind it on the same stream): thread t of 256 adds wave_sum[w], w = t (mod 256),
//     in ascending order, a halving tree joins the 256 partials - the documented tree of
//     tests/test_gpu_device_surface.py::_stats_order, bit for bit what the single-workgroup kernel of round 4 produced.
//     Up to 1 024 waves of rewards (65 536 spacecraft) the same additions are made by ONE wave without LDS or barriers
//     (stats_join1_kernel, round 6: 2.6 - 2.7 us added per step at 65 536 spacecraft instead of 3.5 - 3.7).
// Measured and rejected (profiles/r05/rejected/stats_forms.txt): ONE launch whose last workgroup joins the partials - with
// agent-scope fences 24 us at 65 536 envs and 236 us at 4 Mi (every fence walks the L2), with write-through publications and a
// ticket 7.6 / 99 us; the kernel boundary is the cheap device-wide synchronisation here.
constexpr int STATS_MAX_GRID = 2048;   // workgroups of the first level at most (eight per CU: a 4 Mi batch takes eight trips)
struct StatsScratch {
    double* wave_sum;               // [ceil(n / 64)]
    unsigned* done_part;            // [STATS_MAX_GRID] finished envs per first-level workgroup
};
__global__ __launch_bounds__(256) void stats_kernel(const double* __restrict__ reward, int n,
                                                    const unsigned long long* __restrict__ done_mask, int n_waves, StatsScratch sc) {
    __shared__ unsigned sdone[4];
    const int lane = (int)(threadIdx.x & 63u), hw = (int)(threadIdx.x >> 6);
    const int stride_w = 4 * (int)gridDim.x;
    unsigned nd = 0;
    // four trips at a time: their rewards are loaded together (four 512-byte rows in flight per wave), then summed one by one
    for (int w0 = (int)blockIdx.x * 4 + hw; w0 < n_waves; w0 += 4 * stride_w) {
        double r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = w0 + k * stride_w, i = 64 * w + lane;
            r[k] = (w < n_waves && i < n) ? reward[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = w0 + k * stride_w;
            if (w < n_waves) {                 // (wave-uniform)
                const double ws = wave_sum(r[k]);
                if (lane == 0) {
                    sc.wave_sum[w] = ws;
                    nd += (unsigned)__popcll(done_mask[w]);
                }
            }
        }
    }
    if (lane == 0) sdone[hw] = nd;
    __syncthreads();
    if (threadIdx.x == 0) sc.done_part[blockIdx.x] = sdone[0] + sdone[1] + sdone[2] + sdone[3];
}
// `done_mask` != NULL (the step kernel formed the wave sums itself: bsk_set_step_stats): the done count comes from the waves' ballots.
// 1 024 threads: the first 256 own the reward chains (their order is the documented one and cannot be cut), the other 768 count the
// done bits - integers, any order - beside them, so that the second stream costs the launch no time of its own (one workgroup is
// latency-bound: at 4 Mi spacecraft the masks are another 512 KB).
//
// `seal`: a reset entry point snapshots the last step's scalars and then zeroes the restarted envs' rewards; on a handle whose launches
// live in a HIP graph the host cannot know whether a step has run since, so the device decides: seal_kernel records env 0's counter
// word and episode number behind the reset, every step launch changes one of the two (the step kernel stores cnt[0] on every launch; an
// in-kernel auto-reset zeroes it and increments episodes[0]), and a join that finds them unchanged leaves the snapshot alone.
constexpr int JOIN_THREADS = 1024;
__device__ __forceinline__ bool stats_sealed(const StatsSeal& seal) {
    if (!seal.word || seal.word[2] != 1ull) return false;
    const unsigned long long ep = seal.episodes ? (unsigned long long)(unsigned)seal.episodes[0] : 0ull;
    return seal.word[0] == seal.cnt0[0] && seal.word[1] == ep;
}
__global__ void seal_kernel(StatsSeal seal, unsigned long long* word) {
    word[0] = seal.cnt0[0];
    word[1] = seal.episodes ? (unsigned long long)(unsigned)seal.episodes[0] : 0ull;
    word[2] = 1ull;
}
__global__ __launch_bounds__(JOIN_THREADS) void stats_join_kernel(StatsScratch sc, int n_waves, int n_parts, const unsigned long long* __restrict__ done_mask,
                                                                  double* out_sum, long long* out_done, double* out2, const StatsSeal seal) {
    __shared__ double sr[256];
    __shared__ long long sd[JOIN_THREADS];
    long long nd = 0;
    double acc = 0.0;
    if (threadIdx.x >= 256) {
        constexpr int H = JOIN_THREADS - 256;
        int g = (int)threadIdx.x - 256;
        if (done_mask) {
            for (; g + 15 * H < n_waves; g += 16 * H) {
                unsigned long long m[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) m[k] = done_mask[g + H * k];
#pragma unroll
                for (int k = 0; k < 16; ++k) nd += __popcll(m[k]);
            }
            for (; g < n_waves; g += H) nd += __popcll(done_mask[g]);
        } else {
            for (; g < n_parts; g += H) nd += (long long)sc.done_part[g];
        }
    } else {
        const double* __restrict__ ws = sc.wave_sum;
        int w = (int)threadIdx.x;
        // a row of 256 wave sums per trip (2 KB, coalesced); up to 32 rows in flight (16 waves on the CU: 128 registers each), added in
        // ascending order
        for (; w + 31 * 256 < n_waves; w += 32 * 256) {
            double v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = ws[w + 256 * k];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc += v[k];
        }
        for (; w + 15 * 256 < n_waves; w += 16 * 256) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = ws[w + 256 * k];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k];
        }
        for (; w < n_waves; w += 256) acc += ws[w];
        sr[threadIdx.x] = acc;
    }
    sd[threadIdx.x] = nd;
    __syncthreads();
    for (int off = JOIN_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sd[threadIdx.x] += sd[threadIdx.x + off];
            if (off < 256) sr[threadIdx.x] += sr[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && !stats_sealed(seal)) {
        *out_sum = sr[0];
        *out_done = sd[0];
        out2[0] = sr[0]; out2[1] = (double)sd[0];      // {sum reward, #done} as two doubles: one all-reduce operand
    }
}

// The same join by ONE wave, for batches of up to JOIN1_MAX_WAVES waves (65 536 spacecraft; round 6 - measured at 2 048 waves too, with
// 4 096-wave instantiation: +0.2 us against the workgroup form there, so the limit is where it pays): a 1 024-thread workgroup spends
// most of its 3.7 us starting sixteen waves and walking a ten-level LDS tree with a barrier per level for work that is one memory round
// trip.  Lane l stands for the four "threads" t = l, l + 64, l + 128, l + 192 of the documented order: each chain adds wave_sum[w],
// w = t (mod 256), ascending (rows beyond the batch contribute +0.0, which leaves a sum that started from +0.0 unchanged bit for bit);
// the halving tree's levels 128 and 64 pair values of the SAME lane, its levels 32 ... 1 are shuffles (sr[t] += sr[t + off], t < off) -
// the same additions on the same operands as stats_join_kernel's, no LDS, no barrier; every load of the launch is in flight before
// the first add.  The done count (integers, any order) rides along.
constexpr int JOIN1_MAX_WAVES = 1024;
template <int MAX_WAVES>      // 1 024: 65 536 spacecraft, 16 + 16 loads per lane
__global__ __launch_bounds__(64) void stats_join1_kernel(StatsScratch sc, int n_waves, int n_parts, const unsigned long long* __restrict__ done_mask,
                                                         double* out_sum, long long* out_done, double* out2, const StatsSeal seal) {
    constexpr int JOIN1_ROWS = MAX_WAVES / 256;
    const int lane = (int)threadIdx.x;
    const double* __restrict__ ws = sc.wave_sum;
    const int last = n_waves - 1;
    // (every load unconditional, at a clamped index, the selects afterwards: a conditional load is a branch with its own wait, and a
    // hundred of them in a row are a hundred round trips - 16 us measured)
    double v[4][JOIN1_ROWS];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int r = 0; r < JOIN1_ROWS; ++r) v[k][r] = ws[min(lane + 64 * k + 256 * r, last)];
    }
    long long nd = 0;
    if (done_mask) {
        unsigned long long m[MAX_WAVES / 64];
#pragma unroll
        for (int r = 0; r < MAX_WAVES / 64; ++r) m[r] = done_mask[min(lane + 64 * r, last)];
#pragma unroll
        for (int r = 0; r < MAX_WAVES / 64; ++r) nd += (lane + 64 * r <= last) ? __popcll(m[r]) : 0;
    } else {
        for (int g = lane; g < n_parts; g += 64) nd += (long long)sc.done_part[g];
    }
    double p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < JOIN1_ROWS; ++r) acc += (lane + 64 * k + 256 * r <= last) ? v[k][r] : 0.0;
        p[k] = acc;
    }
    p[0] += p[2];            // level 128: sr[t] += sr[t + 128], t < 128
    p[1] += p[3];
    double s = p[0] + p[1];  // level 64
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);                 // levels 32 ... 1 (lanes >= off carry values nobody reads)
        nd += __shfl_down(nd, off, 64);
    }
    if (lane == 0 && !stats_sealed(seal)) {
        *out_sum = s;
        *out_done = nd;
        out2[0] = s; out2[1] = (double)nd;
    }
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

// ---------------------------------------------------------------------------------------------
// On-device initial-condition sampler (row f4).  Philox4x32-10 (Salmon et al. 2011), written out by
// hand: counter (slot, draw, 0, 0), key (seed_lo, seed_hi); every call yields four 32-bit words =
// two 53-bit uniforms, so each pool slot is reproducible independently of every other slot.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// two uniforms in [0, 1) with 53 random bits each (same bit recipe as numpy's random_double)
__device__ __forceinline__ void philox_u2(unsigned slot, unsigned draw, unsigned k0, unsigned k1, double& a, double& b) {
    unsigned w[4];
    philox4x32_10(slot, draw, 0u, 0u, k0, k1, w);
    a = (double)(((unsigned long long)(w[0] >> 5) << 26) | (w[1] >> 6)) * (1.0 / 9007199254740992.0);
    b = (double)(((unsigned long long)(w[2] >> 5) << 26) | (w[3] >> 6)) * (1.0 / 9007199254740992.0);
}

__global__ void sample_pool_kernel(double* __restrict__ pool, int n_pool, int n_rw, unsigned k0, unsigned k1, double mu) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_pool) return;
    const double PI = 3.14159265358979323846, RPM = 2.0 * PI / 60.0;
    double u[20];
#pragma unroll
    for (int d = 0; d < 10; ++d) philox_u2((unsigned)s, (unsigned)d, k0, k1, u[2 * d], u[2 * d + 1]);
    // orbit: sampled_400km (leo_orbit.py:25-40) -> elem2rv
    const double a = 6371.0 * 1000.0 + 500.0 * 1000.0;
    const double e = 0.05 * u[0], inc = PI * u[1] - 0.5 * PI, Om = 2.0 * PI * u[2], om = 2.0 * PI * u[3], f = 2.0 * PI * u[4];
    const double p = a * (1.0 - e * e), r = p / (1.0 + e * cos(f)), th = om + f;
    const double ct = cos(th), st = sin(th), cO = cos(Om), sO = sin(Om), ci = cos(inc), si = sin(inc);
    const double h = sqrt(mu * p), A = st + e * sin(om), B = ct + e * cos(om), mh = -mu / h;
    auto put = [&](int fld, double v) { pool[(int64_t)fld * n_pool + s] = v; };
    put(BSK_F_R + 0, r * (cO * ct - sO * st * ci)); put(BSK_F_R + 1, r * (sO * ct + cO * st * ci)); put(BSK_F_R + 2, r * (st * si));
    put(BSK_F_V + 0, mh * (cO * A + sO * B * ci)); put(BSK_F_V + 1, mh * (sO * A - cO * B * ci)); put(BSK_F_V + 2, mh * (-B * si));
    // attitude: random_tumble(maxSpinRate = 1e-5) (sc_attitudes.py:3-13, ...Simulator.py:124)
    put(BSK_F_SIGMA + 0, u[5]); put(BSK_F_SIGMA + 1, u[6]); put(BSK_F_SIGMA + 2, u[7]);
    put(BSK_F_OMEGA + 0, 1e-5 * (2.0 * u[8] - 1.0)); put(BSK_F_OMEGA + 1, 1e-5 * (2.0 * u[9] - 1.0));
    put(BSK_F_OMEGA + 2, 1e-5 * (2.0 * u[10] - 1.0));
    // wheel speeds U(-800, 800) RPM (...Simulator.py:155)
    for (int k = 0; k < n_rw; ++k) put(BSK_NF_BASE + k, (1600.0 * u[11 + k] - 800.0) * RPM);
    const int T = BSK_NF_BASE + n_rw;
    // disturbance torque 2e-4 * N(0,1)^3 (...Simulator.py:151-152, 295): Box-Muller on (u15,u16), (u17,u18)
    const double r1 = sqrt(-2.0 * log(1.0 - u[15])), r2 = sqrt(-2.0 * log(1.0 - u[17]));
    put(T + BSK_T_LEXT + 0, 2e-4 * r1 * cos(2.0 * PI * u[16]));
    put(T + BSK_T_LEXT + 1, 2e-4 * r1 * sin(2.0 * PI * u[16]));
    put(T + BSK_T_LEXT + 2, 2e-4 * r2 * cos(2.0 * PI * u[18]));
    for (int k = BSK_T_UCMD; k < BSK_NF_TAIL; ++k) put(T + k, 0.0);
    // battery U(8, 20) W h (...Simulator.py:167)
    put(T + BSK_T_CHARGE, (8.0 + 12.0 * u[19]) * 3600.0);
}

// what a reset leaves in the output buffers of env i: the new episode's first observation (the vec env's convention:
// |sigma_BN|, |omega|, |Omega| / limit in rad/s, charge / 3600 / power_max, 1), zero reward / reason / done / return
__device__ __forceinline__ void init_outputs(const ResetOut& ro, const double* __restrict__ st, int64_t stride, int i) {
    const V3 sg = mk(st[(int64_t)(BSK_F_SIGMA + 0) * stride + i], st[(int64_t)(BSK_F_SIGMA + 1) * stride + i], st[(int64_t)(BSK_F_SIGMA + 2) * stride + i]);
    const V3 w = mk(st[(int64_t)(BSK_F_OMEGA + 0) * stride + i], st[(int64_t)(BSK_F_OMEGA + 1) * stride + i], st[(int64_t)(BSK_F_OMEGA + 2) * stride + i]);
    double om2 = 0.0;
    for (int k = 0; k < ro.n_rw; ++k) {
        const double v = st[(int64_t)(BSK_NF_BASE + k) * stride + i];
        om2 = fma(v, v, om2);
    }
    const double o[5] = {sqrt_nr(dot(sg, sg)), sqrt_nr(dot(w, w)), sqrt_nr(om2) * ro.inv_wheel_limit,
                         st[(int64_t)(BSK_NF_BASE + ro.n_rw + BSK_T_CHARGE) * stride + i] * ro.charge_scale, 1.0};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        ro.obs[(int64_t)k * ro.ostride + i] = o[k];
        if (ro.obs_rm) ro.obs_rm[(int64_t)i * 5 + k] = o[k];
    }
    ro.reward[i] = 0.0;
    ro.reason[i] = 0;
    if (ro.done) ro.done[i] = 0;
    if (ro.ep_return) ro.ep_return[i] = 0.0;
}

// (re)start envs from the pool with the slot rule of the step kernel's auto-reset
__global__ void reset_from_pool_kernel(double* __restrict__ st, int64_t stride, int nf, const double* __restrict__ pool,
                                       int n_pool, const unsigned char* __restrict__ mask, int n, int2* __restrict__ cnt,
                                       int* __restrict__ episodes, unsigned env_base, const ResetOut ro) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    const int ep = episodes[i];
    episodes[i] = ep + 1;
    const unsigned slot = (((unsigned)i + env_base) * 2654435761u + (unsigned)ep * 40503u + 12345u) % (unsigned)n_pool;
    for (int f = 0; f < nf; ++f) st[f * stride + i] = pool[(int64_t)f * n_pool + slot];
    cnt[i] = make_int2(0, 0);
    init_outputs(ro, st, stride, i);
}

// after a reset from host initial conditions: all n envs (idx == NULL) or the m listed ones
__global__ void init_outputs_kernel(const double* __restrict__ st, int64_t stride, const int* __restrict__ idx, int m, const ResetOut ro) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    init_outputs(ro, st, stride, idx ? idx[t] : t);
}

hipError_t launch_sample_pool(double* pool, int n_pool, int n_rw, unsigned long long seed, double mu, hipStream_t s) {
    hipLaunchKernelGGL(sample_pool_kernel, dim3((n_pool + 255) / 256), dim3(256), 0, s, pool, n_pool, n_rw,
                       (unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32), mu);
    return hipGetLastError();
}

hipError_t launch_reset_from_pool(double* st, int64_t stride, int nf, const double* pool, int n_pool, const unsigned char* mask,
                                  int n, int2* cnt, int* episodes, unsigned env_base, const ResetOut& ro, hipStream_t s) {
    hipLaunchKernelGGL(reset_from_pool_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, stride, nf, pool, n_pool, mask, n, cnt,
                       episodes, env_base, ro);
    return hipGetLastError();
}

hipError_t launch_init_outputs(const double* st, int64_t stride, const int* idx, int m, const ResetOut& ro, hipStream_t s) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(init_outputs_kernel, dim3((m + 255) / 256), dim3(256), 0, s, st, stride, idx, m, ro);
    return hipGetLastError();
}

hipError_t launch_seal(const StatsSeal& seal, hipStream_t s) {
    hipLaunchKernelGGL(seal_kernel, dim3(1), dim3(1), 0, s, seal, const_cast<unsigned long long*>(seal.word));
    return hipGetLastError();
}

hipError_t launch_stats(const double* reward, int n, const unsigned long long* done_mask, int n_waves, double* wsum,
                        unsigned* done_part, double* out_sum, long long* out_done, double* out2, bool have_wave_sums, const StatsSeal& seal,
                        hipStream_t s) {
    const bool one_wave = n_waves <= JOIN1_MAX_WAVES;
    if (have_wave_sums) {     // the step kernel wrote wave_sum[] itself: the second level alone
        if (one_wave) hipLaunchKernelGGL(stats_join1_kernel<JOIN1_MAX_WAVES>, dim3(1), dim3(64), 0, s, StatsScratch{wsum, done_part}, n_waves, 0, done_mask, out_sum, out_done, out2, seal);
        else hipLaunchKernelGGL(stats_join_kernel, dim3(1), dim3(JOIN_THREADS), 0, s, StatsScratch{wsum, done_part}, n_waves, 0, done_mask, out_sum, out_done, out2, seal);
        return hipGetLastError();
    }
    // one 256-thread workgroup per four waves of rewards, at most STATS_MAX_GRID of them
    const int grid = std::max(1, std::min((n_waves + 3) / 4, STATS_MAX_GRID));
    hipLaunchKernelGGL(stats_kernel, dim3(grid), dim3(256), 0, s, reward, n, done_mask, n_waves, StatsScratch{wsum, done_part});
    if (one_wave) hipLaunchKernelGGL(stats_join1_kernel<JOIN1_MAX_WAVES>, dim3(1), dim3(64), 0, s, StatsScratch{wsum, done_part}, n_waves, grid, (const unsigned long long*)nullptr,
                                          out_sum, out_done, out2, seal);
    else hipLaunchKernelGGL(stats_join_kernel, dim3(1), dim3(JOIN_THREADS), 0, s, StatsScratch{wsum, done_part}, n_waves, grid, (const unsigned long long*)nullptr,
                            out_sum, out_done, out2, seal);
    return hipGetLastError();
}

// One row of a rollout's history from the handle's output buffers (bsk_step_n where the env steps are separate launches): what
// bsk_get_obs would return now, left in row t of f64[T][5][n] / f64[T][n] / u8[T][n] (any of them may be NULL).
__global__ void hist_row_kernel(const double* __restrict__ obs, const double* __restrict__ reward, const unsigned char* __restrict__ reason,
                                int64_t stride, int n, double* __restrict__ obs_row, double* __restrict__ reward_row,
                                unsigned char* __restrict__ reason_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (obs_row) {
#pragma unroll
        for (int k = 0; k < 5; ++k) obs_row[(int64_t)k * n + i] = obs[(int64_t)k * stride + i];
    }
    if (reward_row) reward_row[i] = reward[i];
    if (reason_row) reason_row[i] = reason[i];
}
hipError_t launch_hist_row(const double* obs, const double* reward, const unsigned char* reason, int64_t stride, int n, double* obs_row,
                           double* reward_row, unsigned char* reason_row, hipStream_t s) {
    if (!obs_row && !reward_row && !reason_row) return hipSuccess;
    hipLaunchKernelGGL(hist_row_kernel, dim3((n + 255) / 256), dim3(256), 0, s, obs, reward, reason, stride, n, obs_row, reward_row, reason_row);
    return hipGetLastError();
}

int stats_done_parts() { return STATS_MAX_GRID; }

hipError_t launch_scatter_reset(double* st, int64_t stride, int nf, const double* ic, const int* idx, int m, int2* cnt,
                                hipStream_t s) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_reset_kernel, dim3((m + 255) / 256), dim3(256), 0, s, st, stride, nf, ic, idx, m, cnt);
    return hipGetLastError();
}

}  // namespace bsk
