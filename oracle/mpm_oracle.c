/*
 * CPU oracle for the MLS-MPM step of noahschwad/flow -- TEST INFRASTRUCTURE ONLY.
 * See mpm_oracle.h for scope, conventions and the "parity unpinned" statement.
 * Every function cites the reference lines it restates; paths are relative to the
 * reference tree, SIM = src/mls-mpm/mlsMpmSimulator.js.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 */
#include "mpm_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void orc_set_threads(int threads) { g_threads = threads < 1 ? 1 : threads; }
int orc_get_threads(void) { return g_threads; }

/* SIM:43 fixedPointMultiplier = 1e7 (exactly representable in f32). */
#define FIXED_POINT_MULTIPLIER 1.0e7f

/* WGSL i32(f32): truncation toward zero, saturating; NaN -> 0. */
static inline int32_t f32_to_i32_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

/* SIM:279-281 encodeFixedPoint */
static inline int32_t encode_fixed_point(float f)
{
    return f32_to_i32_sat(f * FIXED_POINT_MULTIPLIER);
}

/* SIM:282-284 decodeFixedPoint */
static inline float decode_fixed_point(int32_t i)
{
    return (float)i / FIXED_POINT_MULTIPLIER;
}

/* SIM:286-290 getCellPtr */
static inline int32_t cell_ptr(int ix, int iy, int iz, const orc_uniforms *u)
{
    return ix * u->grid_y * u->grid_z + iy * u->grid_z + iz;
}

/* WGSL atomicAdd on atomic<i32>: wraps modulo 2^32. */
static inline void grid_add(int32_t *p, int32_t v)
{
    if (g_threads > 1) {
        __atomic_fetch_add((uint32_t *)p, (uint32_t)v, __ATOMIC_RELAXED);
    } else {
        *(uint32_t *)p += (uint32_t)v;
    }
}

static inline float fractf(float x) { return x - floorf(x); }

/* Quadratic B-spline weights, inlined three times in the reference:
 * SIM:307-312, 347-352, 1322-1328.  w[k][axis]. */
static inline void bspline(const float pos[3], int ci[3], float w[3][3])
{
    for (int a = 0; a < 3; a++) {
        ci[a] = f32_to_i32_sat(pos[a]) - 1;
        float d = fractf(pos[a]) - 0.5f;
        w[0][a] = (0.5f * (0.5f - d)) * (0.5f - d);
        w[1][a] = 0.75f - d * d;
        w[2][a] = (0.5f * (0.5f + d)) * (0.5f + d);
    }
}

/* SIM:262-277 */
void orc_clear_grid(int32_t *cell, float *cellf, int64_t cell_count)
{
    memset(cell, 0, (size_t)cell_count * 4 * sizeof(int32_t));
    memset(cellf, 0, (size_t)cell_count * 4 * sizeof(float));
}

/* SIM:295-333 */
static void p2g1_one(const float *P, int32_t *cell, const orc_uniforms *u)
{
    const float *pos = P + ORC_POS, *vel = P + ORC_VEL, *C = P + ORC_C;
    int ci[3];
    float w[3][3];
    bspline(pos, ci, w);
    for (int gx = 0; gx < 3; gx++)
        for (int gy = 0; gy < 3; gy++)
            for (int gz = 0; gz < 3; gz++) {
                float weight = (w[gx][0] * w[gy][1]) * w[gz][2];
                int cx = ci[0] + gx, cy = ci[1] + gy, cz = ci[2] + gz;
                float d[3] = { ((float)cx + 0.5f) - pos[0],
                               ((float)cy + 0.5f) - pos[1],
                               ((float)cz + 0.5f) - pos[2] };
                float vc[3];
                for (int r = 0; r < 3; r++) {
                    float q = (C[0 + r] * d[0] + C[4 + r] * d[1]) + C[8 + r] * d[2];
                    vc[r] = weight * (vel[r] + q);
                }
                int32_t *c = cell + 4 * (int64_t)cell_ptr(cx, cy, cz, u);
                grid_add(c + 0, encode_fixed_point(vc[0]));
                grid_add(c + 1, encode_fixed_point(vc[1]));
                grid_add(c + 2, encode_fixed_point(vc[2]));
                grid_add(c + 3, encode_fixed_point(weight));
            }
}

void orc_p2g1(const float *particles, int32_t *cell, const orc_uniforms *u)
{
    int64_t n = u->num_particles;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) p2g1_one(particles + i * ORC_STRIDE, cell, u);
}

/* SIM:336-392 */
static void p2g2_one(float *P, int32_t *cell, const orc_uniforms *u)
{
    const float *pos = P + ORC_POS, *C = P + ORC_C;
    int ci[3];
    float w[3][3];
    bspline(pos, ci, w);

    float density = 0.0f;
    for (int gx = 0; gx < 3; gx++)
        for (int gy = 0; gy < 3; gy++)
            for (int gz = 0; gz < 3; gz++) {
                float weight = (w[gx][0] * w[gy][1]) * w[gz][2];
                const int32_t *c = cell + 4 * (int64_t)cell_ptr(ci[0] + gx, ci[1] + gy, ci[2] + gz, u);
                density += decode_fixed_point(c[3]) * weight;
            }
    /* SIM:365-366 densityStore = mix(densityStore, density, 0.05) */
    P[ORC_DENSITY] = P[ORC_DENSITY] * (1.0f - 0.05f) + density * 0.05f;

    float volume = 1.0f / density;
    float ratio = density / u->rest_density;
    float pow5 = ((ratio * ratio) * (ratio * ratio)) * ratio;
    float pressure = fmaxf(0.0f, (pow5 - 1.0f) * u->stiffness);

    /* stress[col][row]; SIM:370-375 */
    float eq16[3][3];
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
            float s = (c == r) ? -pressure : 0.0f;
            float strain = C[4 * c + r] + C[4 * r + c];
            s = s + strain * u->dynamic_viscosity;
            eq16[c][r] = ((volume * -4.0f) * s) * u->dt;
        }

    for (int gx = 0; gx < 3; gx++)
        for (int gy = 0; gy < 3; gy++)
            for (int gz = 0; gz < 3; gz++) {
                float weight = (w[gx][0] * w[gy][1]) * w[gz][2];
                int cx = ci[0] + gx, cy = ci[1] + gy, cz = ci[2] + gz;
                float d[3] = { ((float)cx + 0.5f) - pos[0],
                               ((float)cy + 0.5f) - pos[1],
                               ((float)cz + 0.5f) - pos[2] };
                int32_t *c = cell + 4 * (int64_t)cell_ptr(cx, cy, cz, u);
                for (int r = 0; r < 3; r++) {
                    float m = ((eq16[0][r] * weight) * d[0] + (eq16[1][r] * weight) * d[1])
                              + (eq16[2][r] * weight) * d[2];
                    grid_add(c + r, encode_fixed_point(m));
                }
            }
}

void orc_p2g2(float *particles, int32_t *cell, const orc_uniforms *u)
{
    int64_t n = u->num_particles;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) p2g2_one(particles + i * ORC_STRIDE, cell, u);
}

/* SIM:395-429 */
void orc_update_grid(const int32_t *cell, float *cellf, const orc_uniforms *u)
{
    int64_t count = (int64_t)u->grid_x * u->grid_y * u->grid_z;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < count; i++) {
        const int32_t *c = cell + 4 * i;
        float mass = decode_fixed_point(c[3]);
        if (mass <= 0.0f) continue;
        float vx = decode_fixed_point(c[0]) / mass;
        float vy = decode_fixed_point(c[1]) / mass;
        float vz = decode_fixed_point(c[2]) / mass;
        int32_t idx = (int32_t)i;
        int x = idx / u->grid_z / u->grid_y;
        int y = idx / u->grid_z % u->grid_y;
        int z = idx % u->grid_z;
        if (x < 2 || x > u->grid_x - 2) vx = 0.0f;
        if (y < 2 || y > u->grid_y - 2) vy = 0.0f;
        if (z < 2 || z > u->grid_z - 2) vz = 0.0f;
        float *f = cellf + 4 * i;
        f[0] = vx; f[1] = vy; f[2] = vz; f[3] = mass;
    }
}

/* src/common/noise.js:3-13 */
static inline float tri(float x) { return fabsf(fractf(x) - 0.5f); }

/* src/common/noise.js:27-37 */
static inline void tri3(const float p[3], float out[3])
{
    out[0] = tri(p[2] + tri(p[1] * 1.0f));
    out[1] = tri(p[2] + tri(p[0] * 1.0f));
    out[2] = tri(p[1] + tri(p[0] * 1.0f));
}

/* src/common/noise.js:49-80.  Argument names follow the reference signature
 * (position, speed, time); the g2p call site passes the clock as `speed`. */
void orc_tri_noise3d(const float position[3], float speed, float time, float out[3])
{
    float p[3] = { position[0], position[1], position[2] };
    float bp[3] = { p[0], p[1], p[2] };
    float z = 1.4f;
    float rz[3] = { 0.0f, 0.0f, 0.0f };
    for (int it = 0; it < 4; it++) {           /* float counter 0 <= i <= 3 */
        float b2[3] = { bp[0] * 2.0f, bp[1] * 2.0f, bp[2] * 2.0f };
        float dg[3];
        tri3(b2, dg);
        float s = time * (0.1f * speed);
        for (int a = 0; a < 3; a++) p[a] = p[a] + (dg[a] + s);
        for (int a = 0; a < 3; a++) bp[a] = bp[a] * 1.8f;
        z = z * 1.5f;
        for (int a = 0; a < 3; a++) p[a] = p[a] * 1.2f;
        /* t = trivec(p.zxy + trivec(p.xyz + trivec(p.yzx))) */
        float in0 = tri(p[1]), in1 = tri(p[2]), in2 = tri(p[0]);
        float m0 = tri(p[0] + in0), m1 = tri(p[1] + in1), m2 = tri(p[2] + in2);
        float t0 = tri(p[2] + m0), t1 = tri(p[0] + m1), t2 = tri(p[1] + m2);
        rz[0] = rz[0] + t0 / z;
        rz[1] = rz[1] + t1 / z;
        rz[2] = rz[2] + t2 / z;
        for (int a = 0; a < 3; a++) bp[a] = bp[a] + 0.14f;
    }
    out[0] = rz[0]; out[1] = rz[1]; out[2] = rz[2];
}

static inline void normalize3(float v[3])
{
    float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len;
}

/* SIM:431-443, 484-494, 1316-1362, 1515-1533 (chaos path: gridMode = 0,
 * frontGravityMode = 0, hidePercentage = 0; every other mode uniform is 0). */
static void g2p_one(float *P, const float *cellf, const orc_uniforms *u)
{
    float particle_mass = P[ORC_MASS];
    float pos[3] = { P[ORC_POS], P[ORC_POS + 1], P[ORC_POS + 2] };
    float v[3] = { 0.0f, 0.0f, 0.0f };
    const float gs[3] = { (float)u->grid_x, (float)u->grid_y, (float)u->grid_z };
    const int gsi[3] = { u->grid_x, u->grid_y, u->grid_z };

    /* SIM:490-492 centre gravity */
    {
        float pn[3];
        for (int a = 0; a < 3; a++) pn[a] = pos[a] / (float)(gsi[a] - 1) - 0.5f;
        normalize3(pn);
        for (int a = 0; a < 3; a++) v[a] = v[a] - (pn[a] * 0.1f) * u->dt;
    }
    /* SIM:1317-1320 noise */
    {
        float np[3] = { pos[0] * 0.015f, pos[1] * 0.015f, pos[2] * 0.015f };
        float n[3];
        orc_tri_noise3d(np, u->time, 0.11f, n);
        for (int a = 0; a < 3; a++) n[a] = n[a] - 0.285f;
        normalize3(n);
        for (int a = 0; a < 3; a++) n[a] = n[a] * 0.28f;
        for (int a = 0; a < 3; a++) v[a] = v[a] - (n[a] * u->noise) * u->dt;
    }

    /* SIM:1322-1352 gather */
    int ci[3];
    float w[3][3];
    bspline(pos, ci, w);
    float B[3][3] = { { 0 } };   /* B[col][row] */
    for (int gx = 0; gx < 3; gx++)
        for (int gy = 0; gy < 3; gy++)
            for (int gz = 0; gz < 3; gz++) {
                float weight = (w[gx][0] * w[gy][1]) * w[gz][2];
                int cx = ci[0] + gx, cy = ci[1] + gy, cz = ci[2] + gz;
                float d[3] = { ((float)cx + 0.5f) - pos[0],
                               ((float)cy + 0.5f) - pos[1],
                               ((float)cz + 0.5f) - pos[2] };
                const float *f = cellf + 4 * (int64_t)cell_ptr(cx, cy, cz, u);
                float wv[3] = { f[0] * weight, f[1] * weight, f[2] * weight };
                for (int c = 0; c < 3; c++)
                    for (int r = 0; r < 3; r++) B[c][r] = B[c][r] + wv[r] * d[c];
                for (int r = 0; r < 3; r++) v[r] = v[r] + wv[r];
            }

    /* SIM:1354-1360 mouse force */
    if (u->cursor_interaction == 1u) {
        const float *rd = u->mouse_ray_direction, *ro = u->mouse_ray_origin;
        float b[3] = { pos[0] * 1.0f - ro[0], pos[1] * 1.0f - ro[1], pos[2] * 0.4f - ro[2] };
        float cr[3] = { rd[1] * b[2] - rd[2] * b[1],
                        rd[2] * b[0] - rd[0] * b[2],
                        rd[0] * b[1] - rd[1] * b[0] };
        float dist = sqrtf((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        float f = fmaxf(1.0f - dist * 0.1f, 0.0f);
        float force = f * f;
        for (int a = 0; a < 3; a++) v[a] = v[a] + (u->mouse_force[a] * 1.0f) * force;
        for (int a = 0; a < 3; a++) v[a] = v[a] * particle_mass;
    }

    /* SIM:1362 C = B * 4 */
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) P[ORC_C + 4 * c + r] = B[c][r] * 4.0f;

    /* SIM:1515-1516 */
    for (int a = 0; a < 3; a++) {
        pos[a] = pos[a] + v[a] * u->dt;
        pos[a] = fminf(fmaxf(pos[a], 2.0f), (float)(gsi[a] - 2));
    }
    /* SIM:1518-1527 wall spring */
    for (int a = 0; a < 3; a++) {
        float xn = pos[a] + (v[a] * u->dt) * 3.0f;
        float wall_min = 3.0f, wall_max = gs[a] - 3.0f;
        if (xn < wall_min) v[a] = v[a] + (wall_min - xn) * 0.3f;
        if (xn > wall_max) v[a] = v[a] + (wall_max - xn) * 0.3f;
    }
    /* SIM:1529-1533 */
    for (int a = 0; a < 3; a++) {
        P[ORC_POS + a] = pos[a];
        P[ORC_VEL + a] = v[a];
        P[ORC_DIR + a] = P[ORC_DIR + a] * (1.0f - 0.1f) + v[a] * 0.1f;
    }
}

void orc_g2p(float *particles, const float *cellf, const orc_uniforms *u)
{
    int64_t n = u->num_particles;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) g2p_one(particles + i * ORC_STRIDE, cellf, u);
}

/* SIM:2538-2539 */
void orc_step(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u)
{
    int64_t count = (int64_t)u->grid_x * u->grid_y * u->grid_z;
    orc_clear_grid(cell, cellf, count);
    orc_p2g1(particles, cell, u);
    orc_p2g2(particles, cell, u);
    orc_update_grid(cell, cellf, u);
    orc_g2p(particles, cellf, u);
}

void orc_run(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u,
             int n_steps, double dtime)
{
    orc_uniforms cur = *u;
    for (int k = 0; k < n_steps; k++) {
        cur.time = (float)((double)u->time + (double)k * dtime);
        orc_step(particles, cell, cellf, &cur);
    }
}
