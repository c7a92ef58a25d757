/* Plain-C consumer of bsk_step_n (include/bskgpu.h): what the reference's own main does - a whole run under ONE action
 * (basilisk_env/simulators/leoPowerAttitudeSimulator.py:657-694: 360 steps of action 0) - as ONE launch, then the final observation, the
 * state and the batch scalars read back.  The test compares the printed numbers with the same number of single steps taken through
 * the Python binding.  No history buffers here (a C99 program without the HIP runtime has no device allocator): they may be NULL. */
#include <stdio.h>
#include <stdlib.h>

#include "bskgpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bsk_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) return 9;
    const int n = atoi(argv[2]), n_steps = atoi(argv[3]), n_rw = 3;
    bsk_config cfg;
    CHECK(bsk_default_config(&cfg, n_rw, BSK_GRAV_PM));          /* the reference's wiring: point mass, three wheels */
    bsk_handle* h = NULL;
    CHECK(bsk_create(&cfg, n, 0, NULL, &h));
    const int nf = bsk_n_fields(h);
    double* ic = (double*)calloc((size_t)nf * n, sizeof(double));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(ic, sizeof(double), (size_t)nf * n, f) != (size_t)nf * n) { fprintf(stderr, "cannot read ICs\n"); return 4; }
    fclose(f);
    CHECK(bsk_reset(h, NULL, ic));
    CHECK(bsk_step_n(h, NULL, 0, 10, n_steps, NULL, NULL, NULL));          /* n_steps env steps of 1 s (10 sub-steps), action 0, one launch */
    double* obs = (double*)malloc(sizeof(double) * 5 * n);
    double* rew = (double*)malloc(sizeof(double) * n);
    double* st = (double*)malloc(sizeof(double) * nf * n);
    uint8_t* why = (uint8_t*)malloc(n);
    CHECK(bsk_get_obs_state(h, obs, rew, why, st));
    int32_t* steps = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* ticks = (int32_t*)malloc(sizeof(int32_t) * n);
    CHECK(bsk_get_counters(h, steps, ticks));
    double rsum = 0; int64_t ndone = 0;
    CHECK(bsk_get_batch_stats(h, &rsum, &ndone));
    char name[128]; int vgprs = 0, lds = 0, block = 0, grid = 0;
    CHECK(bsk_kernel_info(h, name, 128, &vgprs, &lds, &block, &grid));
    printf("%.17g %.17g %.17g %.17g %.17g %d %d %d %.17g %lld %s\n", obs[0], obs[n], obs[2 * n + (n - 1)], rew[n - 1], st[(size_t)9 * n], (int)why[0],
           steps[n - 1], ticks[0], rsum, (long long)ndone, name);
    /* refused with a clear message where the rollout kernel is not built, and on nonsense arguments */
    if (bsk_step_n(h, NULL, 7, 10, 3, NULL, NULL, NULL) != BSK_EINVAL || bsk_step_n(h, NULL, 0, 10, 0, NULL, NULL, NULL) != BSK_EINVAL) return 5;
    bsk_destroy(h);
    free(ic); free(obs); free(rew); free(st); free(why); free(steps); free(ticks);
    return 0;
}
