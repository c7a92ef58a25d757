// Micro-benchmark: duration of a kernel that does (almost) nothing as a function of launch geometry,
// register allocation and kernarg size — the floor under the headline step kernel (65 536 spacecraft,
// one RK4 step: 7 us of which 4.6 us is this floor).  Timed with dispatch timestamps
// (hipExtLaunchKernelGGL start/stop events), minimum and median over 200 launches.
//   hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Big {
    double v[160];   // 1 280 bytes of kernarg, like StepArgs
};

__global__ void k_empty() {}
__global__ void k_empty_big(Big b, double* out) {
    if (b.v[0] == 12345.678) out[0] = b.v[159];
}
// forces a large VGPR allocation without doing work
__global__ __launch_bounds__(256) void k_regs(double* out, int n) {
    double a[90];
#pragma unroll
    for (int i = 0; i < 90; ++i) a[i] = (double)(threadIdx.x + i);
    for (int j = 0; j < n; ++j) {
#pragma unroll
        for (int i = 0; i < 90; ++i) a[i] = a[i] * 1.0000001 + a[(i + 1) % 90];
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 90; ++i) s += a[i];
    if (n > 1000000) out[threadIdx.x] = s;
}
// touches memory like the step kernel: reads 20 doubles per lane, writes 22 (coalesced SoA), no arithmetic
__global__ void k_stream(const double* __restrict__ in, double* __restrict__ out, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s[20];
#pragma unroll
    for (int f = 0; f < 20; ++f) s[f] = in[(size_t)f * stride + i];
#pragma unroll
    for (int f = 0; f < 20; ++f) out[(size_t)f * stride + i] = s[f] + 1.0;
    out[(size_t)20 * stride + i] = s[0];
    out[(size_t)21 * stride + i] = s[1];
}

// the same with 1 280 bytes of constants every wave needs before its first load (as the step kernel's HotCfg / pointers): by value in the
// kernarg segment - a fresh location per launch, written by the host - against a pointer to a block that stays put in device memory
struct BigArgs {
    const double* in;
    double* out;
    int stride;
    double c[157];
};
__device__ __forceinline__ void stream_body(const BigArgs& a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double k = 0.0;
#pragma unroll
    for (int j = 0; j < 157; j += 13) k += a.c[j];          // (the constants are used: spread over the whole block)
    double s[20];
#pragma unroll
    for (int f = 0; f < 20; ++f) s[f] = a.in[(size_t)f * a.stride + i];
#pragma unroll
    for (int f = 0; f < 20; ++f) a.out[(size_t)f * a.stride + i] = s[f] + k;
    a.out[(size_t)20 * a.stride + i] = s[0];
    a.out[(size_t)21 * a.stride + i] = s[1];
}
__global__ void k_stream_byvalue(const BigArgs a) { stream_body(a); }
__global__ void k_stream_indirect(const BigArgs* __restrict__ p) {
    BigArgs a;
    __builtin_memcpy(&a, (const __attribute__((address_space(4))) void*)(unsigned long long)p, sizeof(BigArgs));      // constant address space: scalar loads
    stream_body(a);
}

template <class F>
void timeit(const char* name, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < 220; ++i) {
        launch(e0, e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 20) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s min %6.2f us  median %6.2f us\n", name, t.front(), t[t.size() / 2]);
}

int main() {
    double *in, *out;
    const int stride = 65536;
    (void)hipMalloc(&in, (size_t)24 * stride * 8);
    (void)hipMalloc(&out, (size_t)24 * stride * 8);
    (void)hipMemset(in, 0, (size_t)24 * stride * 8);
    Big big{};
    char name[128];
    for (int threads : {64, 256}) {
        for (int total : {64, 4096, 16384, 32768, 65536, 131072, 262144}) {
            const int grid = total / threads;
            if (grid < 1) continue;
            snprintf(name, sizeof name, "empty: %6d threads as %5d x %3d", total, grid, threads);
            timeit(name, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, 0, a, b, 0); });
        }
    }
    timeit("empty + 1 280 B kernarg: 65536 threads as 1024 x 64",
           [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty_big, dim3(1024), dim3(64), 0, 0, a, b, 0, big, out); });
    timeit("~184 VGPRs, no work: 65536 threads as 1024 x 64",
           [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_regs, dim3(1024), dim3(64), 0, 0, a, b, 0, out, 0); });
    timeit("~184 VGPRs, no work: 65536 threads as 256 x 256",
           [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_regs, dim3(256), dim3(256), 0, 0, a, b, 0, out, 0); });
    timeit("stream 20 in / 22 out doubles per lane: 1024 x 64",
           [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_stream, dim3(1024), dim3(64), 0, 0, a, b, 0, in, out, stride); });
    timeit("stream 20 in / 22 out doubles per lane: 256 x 256",
           [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, 0, a, b, 0, in, out, stride); });
    // back-to-back pairs: is the second launch of a pair cheaper (no idle-to-busy transition)?
    timeit("empty 1024 x 64, second of a back-to-back pair", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, 0, nullptr, nullptr, 0);
        hipExtLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, 0, a, b, 0);
    });
    timeit("stream 1024 x 64, second of a back-to-back pair", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_stream, dim3(1024), dim3(64), 0, 0, nullptr, nullptr, 0, in, out, stride);
        hipExtLaunchKernelGGL(k_stream, dim3(1024), dim3(64), 0, 0, a, b, 0, in, out, stride);
    });
    // ... and WITHOUT stamps: wall time per launch of 4 000 back-to-back launches on the null stream (what a stepping loop pays per launch;
    // the stamped figures above include the events' own packets)
    auto wall = [&](const char* name, auto&& launch) {
        double best = 1e9;
        for (int burst = 0; burst < 5; ++burst) {
            for (int k = 0; k < 200; ++k) launch();
            hipDeviceSynchronize();
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            for (int k = 0; k < 4000; ++k) launch();
            hipEventRecord(b, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = std::min(best, (double)ms * 1e3 / 4000);
            hipEventDestroy(a); hipEventDestroy(b);
        }
        printf("%-58s wall %6.2f us per launch (un-stamped, back to back)\n", name, best);
    };
    wall("empty 1024 x 64", [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, 0); });
    wall("empty 256 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); });
    wall("empty + 1 280 B kernarg 1024 x 64", [&] { hipLaunchKernelGGL(k_empty_big, dim3(1024), dim3(64), 0, 0, big, out); });
    wall("stream 20 in / 22 out doubles per lane: 1024 x 64", [&] { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(64), 0, 0, in, out, stride); });
    BigArgs ba;
    ba.in = in; ba.out = out; ba.stride = stride;
    for (int j = 0; j < 157; ++j) ba.c[j] = 1e-3 * j;
    BigArgs* dba;
    hipMalloc(&dba, sizeof(BigArgs));
    hipMemcpy(dba, &ba, sizeof(BigArgs), hipMemcpyHostToDevice);
    wall("stream + 1 280 B of constants by value (kernarg)", [&] { hipLaunchKernelGGL(k_stream_byvalue, dim3(1024), dim3(64), 0, 0, ba); });
    wall("stream + the same constants behind a device pointer", [&] { hipLaunchKernelGGL(k_stream_indirect, dim3(1024), dim3(64), 0, 0, dba); });
    wall("stream + 1 280 B of constants by value (kernarg)", [&] { hipLaunchKernelGGL(k_stream_byvalue, dim3(1024), dim3(64), 0, 0, ba); });
    wall("stream + the same constants behind a device pointer", [&] { hipLaunchKernelGGL(k_stream_indirect, dim3(1024), dim3(64), 0, 0, dba); });
    return 0;
}
