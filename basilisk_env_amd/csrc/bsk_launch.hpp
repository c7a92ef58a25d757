// bsk_launch.hpp — kernel argument block and host-side launch entry points (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsk_device.hpp"

namespace bsk {

// Passed by value to step_kernel (kernarg segment -> SGPRs).
struct StepArgs {
    DevCfg c;
    double* st;                    // state slab [n_fields][stride]
    int2* cnt;                     // per env {env steps, RK4 ticks} since reset
    const int* act;                // actions, device
    double* obs;                   // [5][stride]
    double* reward;                // [stride]
    unsigned long long* done_mask; // [stride/64], one 64-bit ballot per wavefront
    unsigned char* reason;         // [stride]
    double* wave_reward;           // [stride/64]
    int64_t stride;
    int n;
    int substeps;
};

hipError_t launch_step(int grav, int nrw, const StepArgs& a, int block, hipStream_t s);
const void* step_kernel_ptr(int grav, int nrw);
hipError_t launch_stats(const double* wave_reward, const unsigned long long* done_mask, int n_waves, double* out_sum,
                        long long* out_done, hipStream_t s);
hipError_t launch_scatter_reset(double* st, int64_t stride, int nf, const double* ic, const int* idx, int m, int2* cnt,
                                hipStream_t s);

}  // namespace bsk
