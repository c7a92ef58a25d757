/* Plain-C consumer of libbskgpu.so through include/bskgpu.h (no C++, no Python, no torch):
 * create -> reset -> step -> read back, and print a few values for the test to compare with the
 * same steps taken through the Python binding. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bskgpu.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bsk_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* ic_path = argc > 1 ? argv[1] : NULL;
    int n = 96, n_rw = 4;
    bsk_config cfg;
    CHECK(bsk_default_config(&cfg, n_rw, BSK_GRAV_PM_J2));
    if (cfg.struct_size != sizeof(bsk_config) || cfg.abi_version != BSK_ABI_VERSION) return 2;
    bsk_handle* h = NULL;
    CHECK(bsk_create(&cfg, n, 0, NULL, &h));
    int nf = bsk_n_fields(h);
    if (nf != BSK_NF_BASE + n_rw + BSK_NF_TAIL) return 3;
    double* ic = (double*)calloc((size_t)nf * n, sizeof(double));
    FILE* f = ic_path ? fopen(ic_path, "rb") : NULL;
    if (!f || fread(ic, sizeof(double), (size_t)nf * n, f) != (size_t)nf * n) { fprintf(stderr, "cannot read ICs\n"); return 4; }
    fclose(f);
    CHECK(bsk_reset(h, NULL, ic));
    int32_t* act = (int32_t*)malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n; ++i) act[i] = i % 3;
    CHECK(bsk_step(h, act, 25));
    CHECK(bsk_step(h, act, 7));
    double* obs = (double*)malloc(sizeof(double) * 5 * n);
    double* rew = (double*)malloc(sizeof(double) * n);
    uint8_t* done = (uint8_t*)malloc(n);
    CHECK(bsk_get_obs(h, obs, rew, done, NULL));
    double rsum = 0; int64_t ndone = 0;
    CHECK(bsk_get_batch_stats(h, &rsum, &ndone));
    double* st = (double*)malloc(sizeof(double) * nf * n);
    CHECK(bsk_get_state(h, st));
    printf("%.17g %.17g %.17g %.17g %lld\n", obs[0], obs[2 * n + 5], st[0], rsum, (long long)ndone);
    /* error path: NULL arguments are reported, not crashed on */
    if (bsk_step(h, NULL, 1) != BSK_EINVAL || strlen(bsk_last_error()) == 0) return 5;
    bsk_destroy(h);
    free(ic); free(act); free(obs); free(rew); free(done); free(st);
    return 0;
}
