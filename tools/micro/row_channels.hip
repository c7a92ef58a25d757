// Micro-benchmark: how long does ONE wave wait for R rows of 512 bytes loaded at the same lane offset, as a function of the distance
// between the rows?  The step kernel's slab is field-major: a wave's ~46 loads of a launch go to addresses base + f * stride * 8 + 512 w.
// With a power-of-two stride they differ only in high address bits; if the L2 picks its channel from low bits (8 and up) they all queue
// in one channel.  Geometry of the headline launch: 1 024 workgroups of 64 threads, 65 536 columns; every wave times its own batch of
// loads with s_memtime (issue of the first load -> data of the last one), the host reports the mean and median over the waves and the
// kernel's duration.  The slab is written by a previous kernel each round (as the step kernel's is), so the loads come from beyond L2.
// RESULT (profiles/r05/row_channels.txt): the power-of-two stride is the FASTEST here - rows 256 B further apart cost 3 - 4 %, 2 KB
// further 24 % - in all three shapes (loads only; read 46 / write 25 in place; with six output rows, a counter and an action beside the
// slab): the address hash already spreads such rows, and the 2.8 % the PRODUCT kernel gains from the padding (stride_pad_ab.txt) is
// not this effect.
//   hipcc -O3 --offload-arch=gfx950 -o row_channels row_channels.hip && ./row_channels
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int R = 46;
__global__ void k_touch(double* slab, long long stride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        for (int f = 0; f < R; ++f) slab[f * stride + i] = (double)(f + i);
}
__global__ __launch_bounds__(64) void k_rows(const double* __restrict__ slab, long long stride, int n, double* __restrict__ out,
                                             unsigned long long* __restrict__ cyc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double v[R];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int f = 0; f < R; ++f) v[f] = slab[f * stride + i];
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < R; ++f) s += v[f];
    // (the sum depends on every load: its first use is the wait for all of them)
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[i] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the step kernel's shape: read R rows, write the first W of them back in place; launched back to back on one stream (wall time per launch)
constexpr int W = 25;
__global__ __launch_bounds__(64) void k_rw(double* __restrict__ slab, long long stride, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double v[R];
#pragma unroll
    for (int f = 0; f < R; ++f) v[f] = slab[f * stride + i];
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < R; ++f) s += v[f];
    s *= 1e-9;
#pragma unroll
    for (int f = 0; f < W; ++f) slab[f * stride + i] = v[f] * 0.999 + s;
}

// ... and with the launch's other per-env arrays beside the slab: six output rows at the UNPADDED distance (observations + reward), an
// 8-byte counter and a 4-byte action per lane - `obase` shifts the output block's base address by that many bytes
__global__ __launch_bounds__(64) void k_rw2(double* __restrict__ slab, long long stride, int n, double* __restrict__ outp, long long ostride,
                                            unsigned long long* __restrict__ cnt, const int* __restrict__ act) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double v[R];
#pragma unroll
    for (int f = 0; f < R; ++f) v[f] = slab[f * stride + i];
    const unsigned long long c = cnt[i];
    const int a = act[i];
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < R; ++f) s += v[f];
    s = s * 1e-9 + (double)a;
#pragma unroll
    for (int f = 0; f < W; ++f) slab[f * stride + i] = v[f] * 0.999 + s;
#pragma unroll
    for (int k = 0; k < 6; ++k) outp[k * ostride + i] = s + k;
    cnt[i] = c + 1;
}

int main() {
    const int n = 65536;
    const int pads[] = {0, 32, 64, 96, 128, 256, 512, 544, 4096, 4128};
    hipStream_t st;
    hipStreamCreate(&st);
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, n * sizeof(double));
    hipMalloc(&cyc, (n / 64) * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%d rows of 512 B per wave, 1 024 waves (one per SIMD); stride = 65 536 + pad elements\n", R);
    for (int pad : pads) {
        const long long stride = n + pad;
        double* slab;
        hipMalloc(&slab, (size_t)R * stride * sizeof(double));
        std::vector<double> med, mean, kus;
        for (int rep = 0; rep < 60; ++rep) {
            hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, slab, stride, n);
            hipExtLaunchKernelGGL(k_rows, dim3(n / 64), dim3(64), 0, st, e0, e1, 0, slab, stride, n, out, cyc);
            hipStreamSynchronize(st);
            if (rep < 10) continue;
            std::vector<unsigned long long> c(n / 64);
            hipMemcpy(c.data(), cyc, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            double s = 0;
            for (auto x : c) s += (double)x;
            med.push_back((double)c[c.size() / 2]);
            mean.push_back(s / c.size());
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            kus.push_back(ms * 1e3);
        }
        auto mid = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        // back-to-back launches of the read-46 / write-25 kernel: wall time per launch, best of five bursts of 4 000
        double best = 1e9;
        for (int burst = 0; burst < 5; ++burst) {
            for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_rw, dim3(n / 64), dim3(64), 0, st, slab, stride, n);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int k = 0; k < 4000; ++k) hipLaunchKernelGGL(k_rw, dim3(n / 64), dim3(64), 0, st, slab, stride, n);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, (double)ms * 1e3 / 4000);
        }
        printf("pad %5d elements (%6d B): loads only: per-wave wait median %7.0f  mean %7.0f counter ticks, kernel %6.2f us;  read 46 / write 25 in place, back to back: %6.3f us per launch\n",
               pad, pad * 8, mid(med), mid(mean), mid(kus), best);
        hipFree(slab);
    }
    printf("\nread 46 / write 25 slab rows + 6 output rows (unpadded distance) + counter + action, back to back, us per launch:\n");
    for (int pad : {0, 32, 64, 544}) {
        for (int obase : {0, 256, 768, 2048}) {
            const long long stride = n + pad;
            double *slab, *outb;
            unsigned long long* cn;
            int* ac;
            hipMalloc(&slab, (size_t)R * stride * sizeof(double));
            hipMalloc(&outb, (size_t)6 * n * sizeof(double) + 4096);
            hipMalloc(&cn, n * sizeof(unsigned long long));
            hipMalloc(&ac, n * sizeof(int));
            hipMemset(slab, 0, (size_t)R * stride * sizeof(double));
            hipMemset(cn, 0, n * sizeof(unsigned long long));
            hipMemset(ac, 0, n * sizeof(int));
            double* outp = (double*)((char*)outb + obase);
            double best = 1e9;
            for (int burst = 0; burst < 5; ++burst) {
                for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_rw2, dim3(n / 64), dim3(64), 0, st, slab, stride, n, outp, (long long)n, cn, ac);
                hipStreamSynchronize(st);
                hipEventRecord(e0, st);
                for (int k = 0; k < 4000; ++k) hipLaunchKernelGGL(k_rw2, dim3(n / 64), dim3(64), 0, st, slab, stride, n, outp, (long long)n, cn, ac);
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, (double)ms * 1e3 / 4000);
            }
            printf("  slab pad %4d elements, output base + %4d B: %6.3f\n", pad, obase, best);
            hipFree(slab); hipFree(outb); hipFree(cn); hipFree(ac);
        }
    }
    return 0;
}
