// bsk_launch.hpp — kernel argument blocks and host-side launch entry points (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "bsk_device.hpp"

namespace bsk {

// Everything the kernel needs only AFTER the RK4 loop.  It is re-read from the kernarg segment
// behind an opaque pointer once the loop is done, so none of it occupies SGPRs across the loop
// (the loop's HotCfg alone nearly fills the 102-SGPR budget).
struct TailArgs {
    ObsCfg obs_cfg;
    double* st;
    int2* cnt;
    double* obs;                   // [5][stride]
    double* reward;                // [stride]
    unsigned long long* done_mask; // [stride/64], one 64-bit ballot per wavefront
    unsigned char* reason;         // [stride]
    int64_t stride;                // of the state slab's field rows
    int64_t ostride;               // of the observation / terminal-observation rows (bsk_capi.hip: the two differ by the slab's padding)
    int n;
    int substeps;
    // device-side auto-reset (n_pool == 0: off)
    const double* pool;            // [n_fields][n_pool]
    double* term_obs;              // [5][stride]
    int* episodes;                 // [stride]
    int n_pool;
    int n_fields;
    int fsw_lag, nav_lag;
    unsigned env_base;             // global index of this handle's env 0 (sharded batches hash the GLOBAL index)
    int static_charge;             // bare levels: obs[3] is a reset-time constant in the buffers and no battery is empty (StepArgs)
    // device-resident surface (BSK_FLAG_EPISODE_STATS / BSK_FLAG_OBS_ROWMAJOR; NULL: off)
    double* ep_return;             // [stride] return of the running episode (the kernel adds this step's reward)
    double* term_return;           // [stride] return / length of the episode that ended at this step (valid where done)
    int* term_len;                 // [stride]
    unsigned char* done;           // [stride] 0 / 1
    double* obs_rm;                // [n][5] row-major copy of the observation
    int* err;                      // device-visible error word of the handle (page-locked host memory; rarely written)
    unsigned long long* dbg;       // [stride/64] one word per wave, written by probe builds only (bsk_probes.hpp)
    double* wave_sum;              // [stride/64] NULL: off.  bsk_set_step_stats: the wave's reward sum in stats_kernel's order, so that
                                   // a request for the batch scalars behind this launch costs the join kernel alone
};

// Passed by value to step_kernel (kernarg segment -> SGPRs).
template <int NRW, bool DIAG>
struct StepArgs {
    HotCfg<NRW, DIAG> hot;
    const ColdCfg* cold;           // device memory
    const double* st;              // state slab [n_fields][stride]
    const int2* cnt;               // per env {env steps | FSW phase << 20, RK4 ticks} since reset
    const int* act;                // actions, device
    int64_t stride;
    int n;
    int substeps;
    int nav_lag, fsw_lag;   // both also in TailArgs (post-loop re-read); here for the FSW block inside the loop
    int pair_shift;         // pair form: the roles of a workgroup's two waves swap with bit `pair_shift` of its index
    int act_shift;          // actions are int32 (1) or the low words of little-endian int64 (0): byte offset = 8 i >> act_shift
    // Bare levels (no power system): the battery charge never changes, so obs[3] = charge / 3600 / power_max is whatever the last
    // reset left in the observation buffers (every reset entry point writes it) and "battery empty" is a property of the initial
    // conditions.  When the host knows that no spacecraft of the batch (nor of the reset pool) started with an empty battery, the
    // kernel neither loads the charge nor stores obs[3]: 16 of the 407 bytes a K = 1 step moves per spacecraft.  0: as before.
    int static_charge, pad2_;
    const double* ep_return;      // BSK_FLAG_EPISODE_STATS: the running episode's return, loaded with the state (NULL: off)
    PowerCfg power;               // read only by FEAT >= FEAT_POWER
    ExtraCfg extra;               // read only by FEAT_FULL
    TailArgs tail;
};

// Host-side, variant-independent description of the hot constants (built once per handle).
struct StepParams {
    double dt, mu, j2k;
    double inertia[9], dmat[9], dinv[9], wmat[9];   // dmat = I_sc - sum Js g g^T, dinv its inverse, wmat = sum Js g g^T
    double gs[BSK_MAX_RW][3], js[BSK_MAX_RW];
    double f_coulomb;
    int32_t fsw_every;
    ObsCfg obs;
    // spherical harmonics
    double req, planet_rate;
    const double* sh_tab;   // device
    int32_t sh_degree;
    int32_t sh_split;       // first Pines column of the second half of the walk
    int32_t sh_bodies, sh_bodies0, sh_bodies1, sh_chunk1;   // DPP stream: bodies (whole / per half), first chunk of half 1
    int sh_form;            // 1 scalar-load stream, 4 DPP broadcast, 5 DPP broadcast over two cooperating waves
    int pair;               // this launch runs the pair form (dynamics wave + FSW / environment wave per 64 spacecraft)
    int pair_shift;
    int tri;                // this launch runs the three-wave form (rotational / FSW + environment / translational wave)
    int feat;               // FEAT_BARE / FEAT_POWER / FEAT_FULL
    int fsw_lag, nav_lag;
    PowerCfg pc;
    ExtraCfg ex;
};

struct StepBuffers {
    const ColdCfg* cold;
    double* st;
    int2* cnt;
    const int* act;
    double* obs;
    double* reward;
    unsigned long long* done_mask;
    unsigned char* reason;
    int64_t stride;
    int64_t ostride;
    int n;
    int substeps;
    const double* pool;
    double* term_obs;
    int* episodes;
    int n_pool;
    int n_fields;
    unsigned env_base;
    int act_shift;
    int static_charge;
    double* ep_return;
    double* term_return;
    int* term_len;
    unsigned char* done;
    double* obs_rm;
    int* err;
    unsigned long long* dbg;
    double* wave_sum;
};

// the loop's constants as the kernels take them (by value in the kernarg segment)
template <int GRAV, int NRW, bool DIAG>
inline void fill_hot(const StepParams& p, HotCfg<NRW, DIAG>& h) {
    h.h = p.dt; h.h2 = 0.5 * p.dt; h.h3 = p.dt / 3.0; h.h6 = p.dt / 6.0;
    h.nmu = -p.mu; h.j2k = p.j2k;
    for (int i = 0; i < (DIAG ? 3 : 9); ++i) {
        h.Dm[i] = DIAG ? p.dmat[4 * i] : p.dmat[i];
        h.Di[i] = DIAG ? p.dinv[4 * i] : p.dinv[i];
        h.W[i] = DIAG ? p.wmat[4 * i] : p.wmat[i];
    }
    for (int i = 0; i < NRW; ++i) {
        for (int k = 0; k < 3; ++k) h.g[i][k] = p.gs[i][k];
        h.js[i] = p.js[i];
        h.ijs[i] = 1.0 / p.js[i];
    }
    h.fc = p.f_coulomb;
    h.fsw_every = p.fsw_every;
    h.sh_degree = p.sh_degree;
    h.sh_split = p.sh_split;
    h.sh_bodies = p.sh_bodies; h.sh_bodies0 = p.sh_bodies0; h.sh_bodies1 = p.sh_bodies1; h.sh_chunk1 = p.sh_chunk1;
    h.pad_ = 0;
    h.sh_tab = p.sh_tab;
    h.mu_over_req = p.mu / p.req;
    h.req = p.req;
    h.inv_req = 1.0 / p.req;
    h.planet_rate = p.planet_rate;
}

hipError_t launch_step(int grav, int nrw, bool diag, int feat, const StepParams& p, const StepBuffers& b, int block,
                       hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
const void* step_kernel_ptr(int grav, int nrw, bool diag, int feat, int sh_form, bool pair, bool tri = false);
bool pair_available(int grav, bool diag, int feat);
bool tri_available(int grav, bool diag, int feat);
}  // namespace bsk
