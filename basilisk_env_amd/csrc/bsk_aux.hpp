// bsk_aux.hpp — host-side entry points of the small kernels around the step kernel (bsk_aux.hip; internal).
// Kept apart from bsk_launch.hpp so that a change here does not rebuild the step kernel's 36+ instantiations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsk {

// what a reset leaves in the output buffers of the envs it restarts (init_outputs_kernel)
struct ResetOut {
    double* obs;                   // [5][ostride]
    int64_t ostride;
    double* obs_rm;                // [n][5] or NULL
    double* reward;                // [stride]
    unsigned char* reason;         // [stride]
    unsigned char* done;           // [stride] or NULL
    double* ep_return;             // [stride] or NULL
    double inv_wheel_limit, charge_scale;
    int n_rw;
};

hipError_t launch_sample_pool(double* pool, int n_pool, int n_rw, unsigned long long seed, double mu, hipStream_t s);
hipError_t launch_reset_from_pool(double* st, int64_t stride, int nf, const double* pool, int n_pool, const unsigned char* mask,
                                  int n, int2* cnt, int* episodes, unsigned env_base, const ResetOut& ro, hipStream_t s);
// first observation [|sigma_BN|, |omega|, |Omega|/limit, charge/3600/power_max, 1], zero reward / reason / done / episode
// return of freshly reset envs: all n (idx == NULL) or the m listed ones
hipError_t launch_init_outputs(const double* st, int64_t stride, const int* idx, int m, const ResetOut& ro, hipStream_t s);
// "no step since the last reset entry point" as the device sees it (bsk_aux.hip: stats_sealed): env 0's counter word and episode
// number as seal_kernel recorded them behind the reset, compared by the join kernel before it stores
struct StatsSeal {
    const unsigned long long* cnt0;     // the handle's counters (env 0's {steps | phase << 20, ticks} word first)
    const int* episodes;                // [n] or NULL (no IC pool staged)
    const unsigned long long* word;     // [3] {cnt0, episodes[0], 1} at the seal; NULL: never sealed
};
hipError_t launch_seal(const StatsSeal& seal, hipStream_t s);
// batch scalars of the last step (stats_kernel + stats_join_kernel): scratch wave_sum f64[n_waves], done_part u32[stats_done_parts()];
// have_wave_sums: the step kernel filled wave_sum[] (StepBuffers::wave_sum) - the join kernel alone
hipError_t launch_stats(const double* reward, int n, const unsigned long long* done_mask, int n_waves, double* wsum,
                        unsigned* done_part, double* out_sum, long long* out_done, double* out2, bool have_wave_sums, const StatsSeal& seal,
                        hipStream_t s);
hipError_t launch_scatter_reset(double* st, int64_t stride, int nf, const double* ic, const int* idx, int m, int2* cnt,
                                hipStream_t s);

// row t of a rollout's history from the handle's output buffers (bsk_step_n at the levels whose env steps stay separate launches)
hipError_t launch_hist_row(const double* obs, const double* reward, const unsigned char* reason, int64_t stride, int n, double* obs_row,
                           double* reward_row, unsigned char* reason_row, hipStream_t s);

// first-level workgroups of stats_kernel at most = entries of the `done_part` scratch
int stats_done_parts();

}  // namespace bsk
