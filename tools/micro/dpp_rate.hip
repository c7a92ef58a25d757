// Micro-benchmark: issue rate of v_fmac_f64 with the DPP row_newbcast control against plain v_fma_f64,
// one wave per SIMD (1024 workgroups of 64 threads), 8 independent accumulators.
//   hipcc -O3 --offload-arch=gfx950 -o dpp_rate dpp_rate.hip && ./dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double t = seed * (threadIdx.x & 15), b = seed * 0.5;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
#define F(acc, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(b));
            F(a0, 0) F(a1, 1) F(a2, 2) F(a3, 3) F(a4, 4) F(a5, 5) F(a6, 6) F(a7, 7)
            F(a0, 8) F(a1, 9) F(a2, 10) F(a3, 11) F(a4, 12) F(a5, 13) F(a6, 14) F(a7, 15)
#undef F
        } else if constexpr (MODE == 1) {
#define F(acc) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(t), "v"(b));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else {   // dependent pairs as in the recursion: mul, mul, fmac_dpp, then 6 dpp sums on its result
#define D(acc, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(a7));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a6) : "v"(b), "v"(a7));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a7) : "v"(b), "v"(a6));
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a7) : "v"(t), "v"(a6));
            D(a0, 2) D(a1, 3) D(a2, 4) D(a3, 5) D(a4, 6) D(a5, 7)
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a6) : "v"(b), "v"(a7));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a7) : "v"(b), "v"(a6));
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(a7) : "v"(t), "v"(a6));
            D(a0, 10) D(a1, 11) D(a2, 12) D(a3, 13) D(a4, 14) D(a5, 15)
#undef D
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int ops_per_iter, int grid = 1024) {
    double* out;
    hipMalloc(&out, 4096 * 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 100000;
    k<MODE><<<grid, 64>>>(out, 100, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 64>>>(out, iters, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %4d  %8.3f ms  %.2f ns per VALU op per SIMD\n", name, grid, ms, ms * 1e6 / ((double)iters * ops_per_iter * (grid / 1024.0)));
    hipFree(out);
}

int main() {
    run<1>("v_fmac_f64 plain", 16);
    run<0>("v_fmac_f64_dpp row_newbcast", 16);
    run<2>("recursion pattern (18 ops)", 18);
    run<1>("v_fmac_f64 plain", 16, 2048);
    run<0>("v_fmac_f64_dpp row_newbcast", 16, 2048);
    run<2>("recursion pattern (18 ops)", 18, 2048);
    run<2>("recursion pattern (18 ops)", 18, 4096);
    return 0;
}
