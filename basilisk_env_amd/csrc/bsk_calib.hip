// bsk_calib.hip — measurement aid of libbskgpu.so (include/bskgpu.h: bsk_calibrate_fp64): the fp64 FMA rate this device
// SUSTAINS (its clocks under a dense fp64 load, on this box, today), so that a roofline fraction against the nominal 78.6
// TFLOP/s can be read beside what the silicon delivers.  Kept in its own translation unit: it is not part of the step
// kernel's sources (bench.py's kernel fingerprint).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/bskgpu.h"

namespace {

// 16 independent accumulators per lane (no dependent-issue stalls), `iters` x 16 v_fmac_f64 per lane in the operand form the
// step kernel's inner loop mostly uses - accumulator in place, one SGPR factor, one VGPR factor; one wave per workgroup,
// waves_per_simd x (4 x CUs) workgroups.
__global__ __launch_bounds__(64) void fp64_fma_kernel(const double* __restrict__ in, double* __restrict__ out, int iters, double a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double acc[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = in[k]; b[k] = in[16 + k] + 1e-9 * (double)(threadIdx.x & 3); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = fma(a, b[k], acc[k]);
    }
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += acc[k];
    out[i] = sum;
}

}  // namespace

extern "C" int bsk_calibrate_fp64(int device_id, int waves_per_simd, int repeats, double* tflops, double* ns_per_fma_per_simd) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || repeats < 1 || repeats > 64) return BSK_EINVAL;
    int ndev = 0, prev = -1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return BSK_ENODEV;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device_id) != hipSuccess) return BSK_EHIP;
    hipDeviceProp_t prop;
    int rc = BSK_OK;
    double *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<float> ms(repeats, 0.f);
    double host_in[32];
    for (int k = 0; k < 16; ++k) { host_in[k] = 1.0 + 0.01 * k; host_in[16 + k] = 1e-7 * (k + 1); }
    const int iters = 60000;
    int grid = 0;
#define CAL_TRY(expr) do { if ((expr) != hipSuccess) { rc = BSK_EHIP; goto done; } } while (0)
    CAL_TRY(hipGetDeviceProperties(&prop, device_id));
    grid = prop.multiProcessorCount * 4 * waves_per_simd;
    CAL_TRY(hipMalloc(&d_in, sizeof host_in));
    CAL_TRY(hipMalloc(&d_out, (size_t)grid * 64 * sizeof(double)));
    CAL_TRY(hipMemcpy(d_in, host_in, sizeof host_in, hipMemcpyHostToDevice));
    CAL_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CAL_TRY(hipEventCreate(&e0));
    CAL_TRY(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) {                                    // clocks settle
        hipExtLaunchKernelGGL(fp64_fma_kernel, dim3(grid), dim3(64), 0, st, nullptr, nullptr, 0, d_in, d_out, iters, 1.0000001);
        CAL_TRY(hipGetLastError());
    }
    for (int r = 0; r < repeats; ++r) {
        hipExtLaunchKernelGGL(fp64_fma_kernel, dim3(grid), dim3(64), 0, st, e0, e1, 0, d_in, d_out, iters, 1.0000001);
        CAL_TRY(hipGetLastError());
        CAL_TRY(hipStreamSynchronize(st));
        CAL_TRY(hipEventElapsedTime(&ms[r], e0, e1));
    }
    {
        std::sort(ms.begin(), ms.end());
        const double sec = ms[repeats / 2] * 1e-3, fmas_per_wave = 16.0 * iters;   // median
        if (tflops) *tflops = fmas_per_wave * 64.0 * 2.0 * grid / sec / 1e12;
        if (ns_per_fma_per_simd) *ns_per_fma_per_simd = sec * 1e9 / (fmas_per_wave * waves_per_simd);
    }
done:
#undef CAL_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
