// bsk_dispatch.hip — the step kernel's instantiations live in eight translation units (bsk_kernels.hip compiled with -DBSK_TU=0..7:
// about one minute with make -j instead of six in one unit); this is the dispatcher that tries them in turn.
#include "bsk_launch.hpp"

namespace bsk {

#define BSK_UNITS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECL(k)                                                                                                             \
    hipError_t launch_step_tu##k(int grav, int nrw, bool diag, int feat, const StepParams& p, const StepBuffers& b, int block, \
                                 hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool* handled);                             \
    const void* step_kernel_ptr_tu##k(int grav, int nrw, bool diag, int feat, int sh_form, bool pair, bool tri, bool* handled);
BSK_UNITS(DECL)
#undef DECL

hipError_t launch_step(int grav, int nrw, bool diag, int feat, const StepParams& p, const StepBuffers& b, int block,
                       hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    bool handled = false;
    hipError_t e;
#define TRY(k) e = launch_step_tu##k(grav, nrw, diag, feat, p, b, block, s, ev0, ev1, &handled); if (handled) return e;
    BSK_UNITS(TRY)
#undef TRY
    return hipErrorInvalidValue;
}

const void* step_kernel_ptr(int grav, int nrw, bool diag, int feat, int sh_form, bool pair, bool tri) {
    bool handled = false;
    const void* f;
#define TRY(k) f = step_kernel_ptr_tu##k(grav, nrw, diag, feat, sh_form, pair, tri, &handled); if (handled) return f;
    BSK_UNITS(TRY)
#undef TRY
    return nullptr;
}

// pair form: the power / full-scenario levels of the point-mass and J2 kernels with a diagonal hub; three-wave form: the full-scenario level of them
bool pair_available(int grav, bool diag, int feat) { return diag && grav != BSK_GRAV_SH && (feat == FEAT_POWER || feat == FEAT_FULL); }
bool tri_available(int grav, bool diag, int feat) { return diag && grav != BSK_GRAV_SH && feat == FEAT_FULL; }

}  // namespace bsk
