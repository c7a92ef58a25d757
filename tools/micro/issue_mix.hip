// Micro-benchmark: what an fp64 DPP FMA stream costs per SIMD when other instruction types are mixed in,
// at one and at two waves per SIMD (1 024 / 2 048 single-wave workgroups on 1 024 SIMDs).
// One "chunk" = the harmonics walk's 18 VALU ops (4 v_mul_f64 + 14 v_fmac_f64_dpp, same dependency pattern);
// the modes add, per chunk:  1: nothing   2: + global_load_dwordx2 into a ring of 8 + s_waitcnt vmcnt(7)
//                            3: + 2 x s_nop 0   4: + s_sub / s_cmp / s_cbranch (never taken)   5: all of them
//   hipcc -O3 --offload-arch=gfx950 -o issue_mix issue_mix.hip && ./issue_mix     (variants: 256-thread workgroups, 256 VGPRs allocated)
#include <hip/hip_runtime.h>
#include <cstdio>

#define FM(acc, q, b, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q), "v"(b));
#define MUL(r, a, b) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));

template <int MODE, int BLOCK = 64, bool BIGV = false>
__global__ __launch_bounds__(BLOCK) void k(const double* __restrict__ tab, double* out, int bodies, double seed) {
    if constexpr (BIGV) asm volatile("" ::: "v255");   // allocate 256 VGPRs like the harmonics kernel (2 waves per SIMD fit)
    double X1 = seed, X2 = seed, Y1 = seed, Y2 = seed, Z1 = seed, Z2 = seed;
    double P = seed, PP = seed * 0.5, m1 = seed, Bp = seed, nrr = -0.5, ur = 0.25;
    const double* p = tab + (threadIdx.x & 15);
    double q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = seed * (j + 1);
    int rem = 1 << 30;
    for (int b = 0; b < bodies; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jp = (j + 7) & 7;
            double B, B2;
            if constexpr (MODE == 2 || MODE == 5) asm volatile("s_waitcnt vmcnt(7)" : "+v"(q[j]));
            MUL(B, nrr, PP)
            FM(X1, q[jp], Bp, 10)
            FM(B, q[j], m1, 0)
            FM(X2, q[jp], Bp, 11) FM(Y1, q[jp], Bp, 12)
            MUL(m1, ur, B)
            FM(Y2, q[jp], Bp, 13) FM(Z1, q[jp], Bp, 14) FM(Z2, q[jp], Bp, 15)
            if constexpr (MODE == 2 || MODE == 5)
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(q[jp]) : "v"(p + 16 * jp));
            if constexpr (MODE == 3 || MODE == 5) asm volatile("s_nop 0");
            MUL(B2, nrr, P)
            FM(X1, q[j], B, 2)
            FM(B2, q[j], m1, 8)
            FM(X2, q[j], B, 3) FM(Y1, q[j], B, 4)
            if constexpr (MODE == 3 || MODE == 5) asm volatile("s_nop 0");
            MUL(m1, ur, B2)
            FM(Y2, q[j], B, 5) FM(Z1, q[j], B, 6) FM(Z2, q[j], B, 7)
            PP = B; P = B2; Bp = B2;
            if constexpr (MODE == 4 || MODE == 5) {
                if (__builtin_expect(--rem == 0, 0)) { X1 = 0; X2 += 1.0; rem = 1 << 30; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");
    double s = X1 + X2 + Y1 + Y2 + Z1 + Z2 + P + PP;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += q[j];
    out[blockIdx.x * BLOCK + threadIdx.x] = s;
}

template <int MODE, int BLOCK = 64, bool BIGV = false>
void run(const char* name, int grid, const double* tab, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int bodies = 4000;
    k<MODE, BLOCK, BIGV><<<grid * 64 / BLOCK, BLOCK>>>(tab, out, 50, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, BLOCK, BIGV><<<grid * 64 / BLOCK, BLOCK>>>(tab, out, bodies, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d  %8.3f ms  %6.1f ns per chunk per SIMD\n", name, grid / 1024, ms,
           ms * 1e6 / ((double)bodies * 8 * (grid / 1024.0)));
}

int main() {
    double *tab, *out;
    (void)hipMalloc(&tab, 1 << 20);
    (void)hipMemset(tab, 0, 1 << 20);
    (void)hipMalloc(&out, 4096 * 64 * 8);
    for (int grid : {1024, 2048}) {
        run<1>("18 VALU", grid, tab, out);
        run<2>("18 VALU + load + waitcnt", grid, tab, out);
        run<3>("18 VALU + 2 s_nop", grid, tab, out);
        run<4>("18 VALU + s_sub/s_cmp/s_cbranch", grid, tab, out);
        run<5>("18 VALU + all", grid, tab, out);
        run<5, 256>("18 VALU + all, 256-thread workgroups", grid, tab, out);
        run<5, 64, true>("18 VALU + all, 256 VGPRs allocated", grid, tab, out);
        run<5, 256, true>("18 VALU + all, 256-thread wg, 256 VGPRs", grid, tab, out);
    }
    return 0;
}
