/* Plain-C stand-in for the binding a maintainer adds to the reference: what replaces
 * leoPowerAttitudeSimulator.py:594-619 per env step - the full reference scenario, ONE 1 800-sub-step launch per action, the
 * observation and the state read back behind one synchronisation - plus the bookkeeping entry points around it
 * (counters, kernel facts, per-launch timing, env base, explicit synchronisation).  Prints what the test compares with the
 * same calls through the Python binding. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bskgpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bsk_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* ic_path = argc > 1 ? argv[1] : NULL;
    const int n = argc > 2 ? atoi(argv[2]) : 1, n_rw = 3;
    bsk_config cfg;
    CHECK(bsk_default_config(&cfg, n_rw, BSK_GRAV_PM));
    cfg.flags |= BSK_FLAG_POWER | BSK_FLAG_SUN_THIRD_BODY | BSK_FLAG_DRAG | BSK_FLAG_DESAT;
    bsk_handle* h = NULL;
    CHECK(bsk_create(&cfg, n, 0, NULL, &h));
    const int nf = bsk_n_fields(h);
    double* ic = (double*)calloc((size_t)nf * n, sizeof(double));
    FILE* f = ic_path ? fopen(ic_path, "rb") : NULL;
    if (!f || fread(ic, sizeof(double), (size_t)nf * n, f) != (size_t)nf * n) { fprintf(stderr, "cannot read ICs\n"); return 4; }
    fclose(f);
    CHECK(bsk_set_env_base(h, 1000));
    CHECK(bsk_reset(h, NULL, ic));
    int32_t* act = (int32_t*)malloc(sizeof(int32_t) * n);
    double* obs = (double*)malloc(sizeof(double) * 5 * n);
    double* rew = (double*)malloc(sizeof(double) * n);
    uint8_t* why = (uint8_t*)malloc(n);
    double* st = (double*)malloc(sizeof(double) * nf * n);
    CHECK(bsk_profile_begin(h, 8));
    const int actions[3] = {0, 2, 1};
    for (int s = 0; s < 3; ++s) {
        for (int i = 0; i < n; ++i) act[i] = actions[s];
        CHECK(bsk_step(h, act, 1800));                                   /* one 180 s env step */
        CHECK(bsk_get_obs_state(h, obs, rew, why, st));                  /* obs + reward + reason + state, one sync */
        printf("%.17g %.17g %.17g %.17g %.17g %.17g %d\n", obs[0], obs[n], obs[2 * n], obs[3 * n], obs[4 * n], rew[0], (int)why[0]);
    }
    double mean_ms = 0; int launches = 0;
    CHECK(bsk_profile_end(h, &mean_ms, &launches));
    char name[128]; int vgprs = 0, lds = 0, block = 0, grid = 0;
    CHECK(bsk_kernel_info(h, name, (int)sizeof name, &vgprs, &lds, &block, &grid));
    int32_t* steps = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* ticks = (int32_t*)malloc(sizeof(int32_t) * n);
    CHECK(bsk_get_counters(h, steps, ticks));
    void* stream = NULL;
    CHECK(bsk_get_stream(h, &stream));
    CHECK(bsk_sync(h));
    printf("%s %d %d %d %d %d %d %d %s\n", name, block, grid, launches, mean_ms > 0.0 && mean_ms < 100.0, (int)steps[0], (int)ticks[0],
           stream != NULL, bsk_version());
    /* error paths: reported, not crashed on */
    if (bsk_get_obs_state(NULL, obs, rew, why, st) != BSK_EINVAL) return 5;
    if (bsk_step(h, act, 0) == BSK_OK && bsk_step(h, act, -3) == BSK_OK) return 6;
    bsk_destroy(h);
    free(ic); free(act); free(obs); free(rew); free(why); free(st); free(steps); free(ticks);
    return 0;
}
