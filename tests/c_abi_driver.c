/*
 * A plain C host of the C ABI in include/flowmpm.h: no Python, no Node, no C++.
 *
 *   c_abi_driver --sizes
 *       prints sizeof of the public structs as a C compiler lays them out, then the sizes the
 *       library reports (flowmpm_abi_sizes); exits 1 if they differ.  Needs no GPU.
 *
 *   c_abi_driver <dir> dt stiffness rest_density viscosity noise time rdx rdy rdz rox roy roz mfx mfy mfz cursor
 *       runs the golden case whose arrays lie in <dir> as raw little-endian files
 *       (particles_in.bin: 28 floats per particle; expected grids and particle states as named
 *       below) through the five kernels one by one (flowmpm_run_kernel) and then three more
 *       steps (flowmpm_step), for the scatter and the tiled algorithm, and memcmp's every result.
 *       Exit status 0 when every comparison holds.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/flowmpm.h"

enum { GX = 16, GY = 16, GZ = 16, N = 512, CELLS = GX * GY * GZ };

static void *read_file(const char *dir, const char *name, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void *p = malloc(bytes);
    if (!p || fread(p, 1, bytes, f) != bytes || fgetc(f) != EOF) { fprintf(stderr, "%s: expected %zu bytes\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}

static int failures = 0;

static void expect_equal(const char *what, int algorithm, const void *got, const void *want, size_t bytes)
{
    if (memcmp(got, want, bytes) == 0) {
        printf("ok   algorithm %d: %s (%zu bytes)\n", algorithm, what, bytes);
        return;
    }
    const uint32_t *a = got, *b = want;
    size_t first = 0, bad = 0;
    for (size_t i = 0; i < bytes / 4; i++)
        if (a[i] != b[i]) { if (!bad) first = i; bad++; }
    printf("FAIL algorithm %d: %s: %zu of %zu words differ, first at %zu (%08x vs %08x)\n", algorithm, what, bad,
           bytes / 4, first, a[first], b[first]);
    failures++;
}

#define CALL(sim, expr)                                                                      \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != FLOWMPM_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, flowmpm_last_error(sim));          \
            exit(3);                                                                         \
        }                                                                                    \
    } while (0)

static int print_sizes(void)
{
    int64_t lib[4] = {0, 0, 0, 0};
    const int64_t mine[4] = {(int64_t)sizeof(flowmpm_config), (int64_t)sizeof(flowmpm_uniforms),
                             (int64_t)sizeof(flowmpm_mode_uniforms), (int64_t)sizeof(flowmpm_timings)};
    if (flowmpm_abi_sizes(lib) != FLOWMPM_OK) return 1;
    printf("header  %lld %lld %lld %lld\n", (long long)mine[0], (long long)mine[1], (long long)mine[2], (long long)mine[3]);
    printf("library %lld %lld %lld %lld\n", (long long)lib[0], (long long)lib[1], (long long)lib[2], (long long)lib[3]);
    return memcmp(lib, mine, sizeof lib) != 0;
}

int main(int argc, char **argv)
{
    if (argc == 2 && strcmp(argv[1], "--sizes") == 0) return print_sizes();
    if (argc != 18) { fprintf(stderr, "usage: %s --sizes | <dir> <16 uniform values>\n", argv[0]); return 2; }
    const char *dir = argv[1];
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = (float)strtod(argv[2 + i], NULL);      /* double -> float, as the other hosts do */

    float *particles_in = read_file(dir, "particles_in.bin", (size_t)N * 28 * 4);
    int32_t *cell_p2g1 = read_file(dir, "cell_after_p2g1.bin", (size_t)CELLS * 16);
    int32_t *cell_p2g2 = read_file(dir, "cell_after_p2g2.bin", (size_t)CELLS * 16);
    float *cellf_update = read_file(dir, "cellf_after_update.bin", (size_t)CELLS * 16);
    float *particles_1 = read_file(dir, "particles_step1.bin", (size_t)N * 28 * 4);
    float *particles_4 = read_file(dir, "particles_step4.bin", (size_t)N * 28 * 4);
    int32_t *cell_4 = read_file(dir, "cell_step4.bin", (size_t)CELLS * 16);

    int32_t *cell = malloc((size_t)CELLS * 16);
    float *cellf = malloc((size_t)CELLS * 16), *particles = malloc((size_t)N * 28 * 4);
    const int algorithms[2] = {FLOWMPM_ALGO_SCATTER, FLOWMPM_ALGO_TILED};
    for (int a = 0; a < 2; a++) {
        flowmpm_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.grid_x = GX; cfg.grid_y = GY; cfg.grid_z = GZ;
        cfg.algorithm = algorithms[a];
        cfg.rebin_interval = 2;
        cfg.max_particles = N;
        flowmpm_sim *sim = NULL;
        CALL(NULL, flowmpm_create(&cfg, &sim));
        CALL(sim, flowmpm_upload_particles(sim, particles_in, NULL, N));
        flowmpm_uniforms u;
        memset(&u, 0, sizeof u);
        u.dt = v[0]; u.stiffness = v[1]; u.rest_density = v[2]; u.dynamic_viscosity = v[3]; u.noise = v[4]; u.time = v[5];
        for (int k = 0; k < 3; k++) { u.mouse_ray_direction[k] = v[6 + k]; u.mouse_ray_origin[k] = v[9 + k]; u.mouse_force[k] = v[12 + k]; }
        u.cursor_interaction = (uint32_t)v[15];
        CALL(sim, flowmpm_set_uniforms(sim, &u));

        /* the five kernels in the reference's submission order (mlsMpmSimulator.js:2538) */
        CALL(sim, flowmpm_run_kernel(sim, FLOWMPM_KERNEL_CLEAR_GRID));
        CALL(sim, flowmpm_run_kernel(sim, FLOWMPM_KERNEL_P2G1));
        CALL(sim, flowmpm_download_grid_i32(sim, cell, CELLS));
        expect_equal("cellBuffer after p2g1", algorithms[a], cell, cell_p2g1, (size_t)CELLS * 16);
        CALL(sim, flowmpm_run_kernel(sim, FLOWMPM_KERNEL_P2G2));
        CALL(sim, flowmpm_download_grid_i32(sim, cell, CELLS));
        expect_equal("cellBuffer after p2g2", algorithms[a], cell, cell_p2g2, (size_t)CELLS * 16);
        CALL(sim, flowmpm_run_kernel(sim, FLOWMPM_KERNEL_UPDATE_GRID));
        CALL(sim, flowmpm_download_grid_f32(sim, cellf, CELLS));
        expect_equal("cellBufferF after updateGrid", algorithms[a], cellf, cellf_update, (size_t)CELLS * 16);
        CALL(sim, flowmpm_run_kernel(sim, FLOWMPM_KERNEL_G2P));
        int64_t count = 0;
        CALL(sim, flowmpm_download_particles(sim, particles, NULL, N, &count));
        if (count != N) { printf("FAIL %lld particles came back\n", (long long)count); failures++; }
        expect_equal("particles after one step", algorithms[a], particles, particles_1, (size_t)N * 28 * 4);

        /* three more frames; the clock of frame k is time + k / 60 */
        CALL(sim, flowmpm_step(sim, 3, 1.0 / 60));
        CALL(sim, flowmpm_download_particles(sim, particles, NULL, N, &count));
        expect_equal("particles after four steps", algorithms[a], particles, particles_4, (size_t)N * 28 * 4);
        CALL(sim, flowmpm_download_grid_i32(sim, cell, CELLS));
        expect_equal("cellBuffer of the fourth step", algorithms[a], cell, cell_4, (size_t)CELLS * 16);
        int64_t sums[4], mass = 0;
        CALL(sim, flowmpm_grid_sums(sim, sums));
        for (int c = 0; c < CELLS; c++) mass += cell_4[4 * c + 3];
        if (sums[3] != mass) { printf("FAIL grid mass sum %lld vs %lld\n", (long long)sums[3], (long long)mass); failures++; }
        CALL(sim, flowmpm_destroy(sim));
    }
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
