// Round-trip time of a value handed between two waves of one workgroup through LDS without a barrier (the three-wave form's
// exchange: rows first, tag last; the consumer reads the tag first and the rows after it, in one batch, and repeats until the
// tag is the expected one).  Wave 0 publishes k, wave 1 answers k, ... N round trips; cycles per round trip = 2 hops.
// Also: the same with W VALU instructions of independent work between a wave's consume and its publish (work hides nothing of
// the hop: it is on the chain), and the one-way latency of a barrier-based hand-over for comparison.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_pingpong tools/micro/lds_pingpong.hip && /tmp/lds_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct X {
    double row[2][3][64];
    int tag[2][64];
};
typedef X __attribute__((address_space(3))) * XP;
__device__ __forceinline__ void st(double __attribute__((address_space(3))) * p, double v) {
    __hip_atomic_store((long long __attribute__((address_space(3)))*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double ld(const double __attribute__((address_space(3))) * p) {
    return __longlong_as_double(__hip_atomic_load((long long __attribute__((address_space(3)))*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

template <int WORK, bool BARRIER>
__global__ __launch_bounds__(128) void pingpong(int n, unsigned long long* out, double* sink) {
    __shared__ X xs;
    XP x = (XP)&xs;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    x->tag[0][lane] = 0; x->tag[1][lane] = 0;
    __syncthreads();
    double a = 1.0 + lane, b = 0.5, c = 0.25;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 1; k <= n; ++k) {
        if (BARRIER) {
            if (w == 0) { st(&x->row[0][0][lane], a); st(&x->row[0][1][lane], b); st(&x->row[0][2][lane], c); }
            __syncthreads();
            if (w == 1) { a = ld(&x->row[0][0][lane]) + 1.0; b = ld(&x->row[0][1][lane]); c = ld(&x->row[0][2][lane]);
                          st(&x->row[1][0][lane], a); st(&x->row[1][1][lane], b); st(&x->row[1][2][lane], c); }
            __syncthreads();
            if (w == 0) { a = ld(&x->row[1][0][lane]); b = ld(&x->row[1][1][lane]); c = ld(&x->row[1][2][lane]); }
        } else {
            if (w == 0) {
                st(&x->row[0][0][lane], a); st(&x->row[0][1][lane], b); st(&x->row[0][2][lane], c);
                __hip_atomic_store(&x->tag[0][lane], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {
                    const int t = __hip_atomic_load(&x->tag[1][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a = ld(&x->row[1][0][lane]); b = ld(&x->row[1][1][lane]); c = ld(&x->row[1][2][lane]);
                    if (__builtin_amdgcn_ballot_w64(t != k) == 0) break;
                }
            } else {
                for (;;) {
                    const int t = __hip_atomic_load(&x->tag[0][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a = ld(&x->row[0][0][lane]); b = ld(&x->row[0][1][lane]); c = ld(&x->row[0][2][lane]);
                    if (__builtin_amdgcn_ballot_w64(t != k) == 0) break;
                }
#pragma unroll
                for (int i = 0; i < WORK; ++i) a = fma(a, 1.0000001, 1e-9);      // a dependent fp64 chain
                st(&x->row[1][0][lane], a); st(&x->row[1][1][lane], b); st(&x->row[1][2][lane], c);
                __hip_atomic_store(&x->tag[1][lane], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 128 + threadIdx.x] = a + b + c;
}

template <int WORK, bool BARRIER>
static void run(const char* name, int blocks) {
    const int n = 20000;
    unsigned long long* out; double* sink;
    hipMalloc(&out, blocks * sizeof(unsigned long long)); hipMalloc(&sink, blocks * 128 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pingpong<WORK, BARRIER>), dim3(blocks), dim3(128), 0, 0, n, out, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pingpong<WORK, BARRIER>), dim3(blocks), dim3(128), 0, 0, n, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    printf("%-34s blocks %4d: %.0f cycles per round trip (2 hops), %.1f ns per round trip, counter %.2f GHz\n", name, blocks, mean / n, ms * 1e6 / n, mean / (ms * 1e6));
    hipFree(out); hipFree(sink);
}

int main() {
    for (int blocks : {1, 128}) {
        run<0, false>("tagged slots, no work", blocks);
        run<16, false>("tagged slots, 16 dependent FMAs", blocks);
        run<64, false>("tagged slots, 64 dependent FMAs", blocks);
        run<0, true>("barrier hand-over", blocks);
    }
    return 0;
}
