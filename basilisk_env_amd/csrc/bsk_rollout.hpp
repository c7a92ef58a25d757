// bsk_rollout.hpp — host-side entry points of the open-loop rollout kernel (bsk_rollout.hip; internal).
#pragma once
#include "bsk_launch.hpp"

namespace bsk {

struct RolloutBuffers {
    const int* actions;            // [T][n] device, or NULL: `const_action` at every step
    double* obs_hist;              // [T][5][n] device (NULL: not recorded)
    double* reward_hist;           // [T][n]
    unsigned char* reason_hist;    // [T][n]
    int n_steps, const_action;
};

// built for: point mass / J2 at the bare level (every wheel set, diagonal and general hub)
bool rollout_available(int grav, int feat);
hipError_t launch_rollout(int grav, int nrw, bool diag, const StepParams& p, const StepBuffers& b, const RolloutBuffers& r, int block,
                          hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
const void* rollout_kernel_ptr(int grav, int nrw, bool diag, bool act);

}  // namespace bsk
