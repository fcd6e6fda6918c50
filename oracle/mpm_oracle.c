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
                    for (int r = 0; r < 3; r++)
                        B[c][r] = B[c][r] + wv[r] * d[c];
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


/* ---- g2p with the mode branches ------------------------------------------------------- */

/* Colours the nine population-segment modes compare against (SIM:766, 847, 906, 965, 1024,
 * 1083, 1142, 1201, 1260), in the order of the mode uniforms. */
static const float k_segment_color[ORC_SEGMENTS][3] = {
    { 1.0f, 0.3372549f, 0.717647f }, { 0.737f, 0.588f, 0.976f }, { 0.773f, 0.941f, 0.949f },
    { 0.0f, 0.455f, 0.529f },        { 0.082f, 0.224f, 0.263f }, { 0.137f, 0.137f, 0.337f },
    { 0.314f, 0.071f, 0.102f },      { 1.0f, 0.392f, 0.145f },   { 0.839f, 0.957f, 0.600f },
};

static inline float length3(const float v[3]) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* boostTime / boostDuration clamped, smoothstep-eased (SIM:751-753 and its copies). */
static inline float eased_boost(float boost_time, float boost_duration)
{
    float bp = clampf(boost_time / boost_duration, 0.0f, 1.0f);
    return (bp * bp) * (3.0f - bp * 2.0f);
}

/* v += (direction * strength) * dt */
static inline void push(float v[3], const float dir[3], float strength, float dt)
{
    for (int a = 0; a < 3; a++) v[a] = v[a] + (dir[a] * strength) * dt;
}

/* "Outside the sphere -> push back in" (SIM:517-528, 535-546, 740-758, 773-794 ...).
 * boosted: strength = (excess * 0.3) * mix(1, 20, eased); plain: strength = excess * 0.3. */
static inline void contain_in_sphere(float v[3], const float pos[3], const float center[3], float radius,
                                     int boosted, float eased, float dt)
{
    float to[3] = { pos[0] - center[0], pos[1] - center[1], pos[2] - center[2] };
    float dist = length3(to);
    if (dist > radius) {
        float toward[3] = { -(to[0] / dist), -(to[1] / dist), -(to[2] / dist) };
        float excess = dist - radius;
        float strength = excess * 0.3f;
        if (boosted) strength = strength * (1.0f * (1.0f - eased) + 20.0f * eased);
        push(v, toward, strength, dt);
    }
}

/* "Inside the negative-space sphere -> push out" (SIM:821-840 and its copies). */
static inline void repel_from_sphere(float v[3], const float pos[3], const float center[3], float radius,
                                     float eased, float dt)
{
    float to[3] = { pos[0] - center[0], pos[1] - center[1], pos[2] - center[2] };
    float dist = length3(to);
    if (dist < radius) {
        float away[3] = { to[0] / dist, to[1] / dist, to[2] / dist };
        float depth = radius - dist;
        float strength = (depth * 0.3f) * (1.0f * (1.0f - eased) + 20.0f * eased);
        push(v, away, strength, dt);
    }
}

static void g2p_modes_one(float *P, int32_t index, const float *cellf, const orc_uniforms *u,
                          const orc_mode_uniforms *m)
{
    float particle_mass = P[ORC_MASS];
    float pos[3] = { P[ORC_POS], P[ORC_POS + 1], P[ORC_POS + 2] };
    const float color[3] = { P[ORC_COLOR], P[ORC_COLOR + 1], P[ORC_COLOR + 2] };
    float v[3] = { 0.0f, 0.0f, 0.0f };
    const float gs[3] = { (float)u->grid_x, (float)u->grid_y, (float)u->grid_z };
    const int gsi[3] = { u->grid_x, u->grid_y, u->grid_z };
    const float dt = u->dt;
    /* SIM:441-442 the first hidePercentage percent of the buffer are "hidden" */
    const float index_percent = ((float)(uint32_t)index / (float)u->num_particles) * 100.0f;
    const int hide = u->hide_percentage > 0.0f && index_percent < u->hide_percentage;

    if (u->grid_mode == 1u && !hide) {
        /* SIM:445-482 attraction to a target on the image grid */
        int32_t gw = (int32_t)m->grid_width;
        int32_t gxi = gw ? index % gw : 0;         /* WGSL: x % 0 = 0, x / 0 = x */
        int32_t gyi = gw ? index / gw : index;
        float hash = ((float)index * 12.9898f + (float)gxi * 78.233f) + (float)gyi * 37.719f;
        float random_x = fractf(hash) * 2.0f - 1.0f;
        float random_y = fractf(hash * 43758.5453f) * 2.0f - 1.0f;
        float offset_x = (random_x * m->grid_spacing_x) * m->grid_random_offset;
        float offset_y = (random_y * m->grid_spacing_y) * m->grid_random_offset;
        float target[3] = { (m->grid_start_x + (float)gxi * m->grid_spacing_x) + offset_x,
                            (m->grid_start_y + (float)gyi * m->grid_spacing_y) + offset_y, m->grid_z };
        float np[3] = { pos[0] * 0.92f, pos[1] * 0.92f, pos[2] * 0.92f }, n[3];
        orc_tri_noise3d(np, 1.15f, u->time, n);
        for (int a = 0; a < 3; a++) target[a] = target[a] + (n[a] * m->grid_noise_strength) * 2.0f;
        float zp[3] = { pos[0] * 0.85f + 100.0f, pos[1] * 0.85f + 200.0f, pos[2] * 0.85f + 300.0f }, zn[3];
        orc_tri_noise3d(zp, 0.02f, u->time, zn);
        target[2] = target[2] + (zn[2] * m->grid_noise_strength) * 2.0f;
        float to[3] = { target[0] - pos[0], target[1] - pos[1], target[2] - pos[2] };
        float dist = length3(to);
        float dir[3] = { to[0] / dist, to[1] / dist, to[2] / dist };
        push(v, dir, dist * 10.0f, dt);
    } else if (u->front_gravity_mode == 1u && !hide) {
        /* SIM:485-488: pull toward -z */
        const float front[3] = { 0.0f, 0.0f, -1.0f };
        for (int a = 0; a < 3; a++) v[a] = v[a] + (front[a] * 0.1f) * dt;
    } else {
        /* SIM:490-492 centre gravity */
        float pn[3];
        for (int a = 0; a < 3; a++) pn[a] = pos[a] / (float)(gsi[a] - 1) - 0.5f;
        normalize3(pn);
        for (int a = 0; a < 3; a++) v[a] = v[a] - (pn[a] * 0.1f) * dt;
    }

    /* SIM:497-511 cylinder repulsion */
    if (m->cylinder_mode == 1u && !hide) {
        float to[3] = { pos[0] - m->cylinder_center_x, pos[1] - m->cylinder_center_y, pos[2] - pos[2] };
        float flat[3] = { to[0], to[1], 0.0f };
        float dist = length3(flat);
        if (dist < m->cylinder_radius) {
            float dir[3] = { flat[0] / dist, flat[1] / dist, flat[2] / dist };
            float strength = (1.0f - dist / m->cylinder_radius) * 0.5f;
            push(v, dir, strength, dt);
        }
    }
    /* SIM:514-529 sphere containment, SIM:532-547 two-colour sphere */
    if (m->sphere_containment_mode == 1u && !hide) contain_in_sphere(v, pos, m->sphere_center, m->sphere_radius, 0, 0.0f, dt);
    if (m->two_color_sphere_mode == 1u && !hide) contain_in_sphere(v, pos, m->sphere_center, m->sphere_radius, 0, 0.0f, dt);

    /* SIM:550-696 polygon containment in the xy plane */
    if (m->polygon_containment_mode == 1u && !hide) {
        const float px = pos[0], py = pos[1];
        int inside = 0;
        const int count = (int)m->polygon_vertex_count;
        for (int i = 0; i < count; i++) {
            int j = (i == count - 1) ? 0 : i + 1;
            const float *v0 = m->polygon_vertices[i < 15 ? i : 15];
            const float *v1 = m->polygon_vertices[j < 15 ? j : 15];
            int above0 = v0[1] > py, above1 = v1[1] > py;
            if (above0 != above1) {
                float t = (py - v0[1]) / (v1[1] - v0[1]);
                float x_intersect = v0[0] + (v1[0] - v0[0]) * t;
                if (x_intersect > px) inside = (inside + 1) % 2;
            }
        }
        if (inside == 0) {
            float min_dist = 9999.0f, nearest[2] = { 0.0f, 0.0f };
            for (int i = 0; i < count; i++) {
                int j = (i == count - 1) ? 0 : i + 1;
                const float *v0 = m->polygon_vertices[i < 15 ? i : 15];
                const float *v1 = m->polygon_vertices[j < 15 ? j : 15];
                float edge[2] = { v1[0] - v0[0], v1[1] - v0[1] };
                float to_p[2] = { px - v0[0], py - v0[1] };
                float edge_len = sqrtf(edge[0] * edge[0] + edge[1] * edge[1]);
                float edge_len_sq = edge_len * edge_len;
                float t = (edge_len_sq < 0.0001f)
                              ? 0.0f
                              : clampf((to_p[0] * edge[0] + to_p[1] * edge[1]) / edge_len_sq, 0.0f, 1.0f);
                float closest[2] = { v0[0] + edge[0] * t, v0[1] + edge[1] * t };
                float dx = px - closest[0], dy = py - closest[1];
                float dist = sqrtf(dx * dx + dy * dy);
                if (dist < min_dist) { min_dist = dist; nearest[0] = closest[0]; nearest[1] = closest[1]; }
            }
            float to[3] = { nearest[0] - pos[0], nearest[1] - pos[1], pos[2] - pos[2] };
            float dist = length3(to);
            float dir[3] = { to[0] / dist, to[1] / dist, to[2] / dist };
            push(v, dir, dist * 0.3f, dt);
        }
    }

    /* SIM:699-759 colour spheres */
    if (m->color_spheres_mode == 1u && !hide) {
        float min_dist = 999.0f;
        int closest = 0;
        for (int i = 0; i < 10; i++) {
            const float *pc = m->palette_colors[i < 9 ? i : 9];
            float d[3] = { color[0] - pc[0], color[1] - pc[1], color[2] - pc[2] };
            float dist = length3(d);
            if (dist < min_dist) { min_dist = dist; closest = i; }
        }
        contain_in_sphere(v, pos, m->color_sphere_positions[closest < 9 ? closest : 9], m->color_sphere_radius, 1,
                          eased_boost(m->color_spheres_boost_time, m->color_spheres_boost_duration), dt);
    }

    /* SIM:762-1313 population segments: main sphere for everyone, a smaller sphere for the
     * segment's colour, and a negative-space sphere that repels */
    for (int k = 0; k < ORC_SEGMENTS; k++) {
        if (m->segment_mode[k] != 1u || hide) continue;
        float cd[3] = { color[0] - k_segment_color[k][0], color[1] - k_segment_color[k][1],
                        color[2] - k_segment_color[k][2] };
        int is_segment = length3(cd) < 0.1f;
        float eased = eased_boost(m->segment_boost_time[k], m->segment_boost_duration[k]);
        contain_in_sphere(v, pos, m->sphere_center, m->segment_main_radius[k], 1, eased, dt);
        if (is_segment)
            contain_in_sphere(v, pos, m->segment_sphere_center[k], m->segment_sphere_radius[k], 1, eased, dt);
        repel_from_sphere(v, pos, m->segment_negative_center[k], m->negative_space_sphere_radius, eased, dt);
    }

    /* SIM:1317-1320 noise, only outside grid and front-gravity mode */
    if (u->grid_mode == 0u && u->front_gravity_mode == 0u) {
        float np[3] = { pos[0] * 0.015f, pos[1] * 0.015f, pos[2] * 0.015f }, n[3];
        orc_tri_noise3d(np, u->time, 0.11f, n);
        for (int a = 0; a < 3; a++) n[a] = n[a] - 0.285f;
        normalize3(n);
        for (int a = 0; a < 3; a++) n[a] = n[a] * 0.28f;
        for (int a = 0; a < 3; a++) v[a] = v[a] - (n[a] * u->noise) * dt;
    }

    /* SIM:1322-1352 gather, skipped in grid mode and for hidden particles */
    float B[3][3] = { { 0 } };
    if (u->grid_mode == 0u && !hide) {
        int ci[3];
        float w[3][3];
        bspline(pos, ci, w);
        for (int gx = 0; gx < 3; gx++)
            for (int gy = 0; gy < 3; gy++)
                for (int gz = 0; gz < 3; gz++) {
                    float weight = (w[gx][0] * w[gy][1]) * w[gz][2];
                    int cx = ci[0] + gx, cy = ci[1] + gy, cz = ci[2] + gz;
                    float d[3] = { ((float)cx + 0.5f) - pos[0], ((float)cy + 0.5f) - pos[1],
                                   ((float)cz + 0.5f) - pos[2] };
                    const float *f = cellf + 4 * (int64_t)cell_ptr(cx, cy, cz, u);
                    float wv[3] = { f[0] * weight, f[1] * weight, f[2] * weight };
                    for (int c = 0; c < 3; c++)
                        for (int r = 0; r < 3; r++) B[c][r] = B[c][r] + wv[r] * d[c];
                    for (int r = 0; r < 3; r++) v[r] = v[r] + wv[r];
                }
    }

    /* SIM:1355-1360 mouse force */
    if (u->grid_mode == 0u && u->cursor_interaction == 1u && !hide) {
        const float *rd = u->mouse_ray_direction, *ro = u->mouse_ray_origin;
        float b[3] = { pos[0] * 1.0f - ro[0], pos[1] * 1.0f - ro[1], pos[2] * 0.4f - ro[2] };
        float cr[3] = { rd[1] * b[2] - rd[2] * b[1], rd[2] * b[0] - rd[0] * b[2], rd[0] * b[1] - rd[1] * b[0] };
        float f = fmaxf(1.0f - length3(cr) * 0.1f, 0.0f);
        float force = f * f;
        for (int a = 0; a < 3; a++) v[a] = v[a] + (u->mouse_force[a] * 1.0f) * force;
        for (int a = 0; a < 3; a++) v[a] = v[a] * particle_mass;
    }

    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) P[ORC_C + 4 * c + r] = B[c][r] * 4.0f;      /* SIM:1362 */

    /* SIM:1365-1476 damping */
    if (u->front_gravity_mode == 1u)
        for (int a = 0; a < 3; a++) v[a] = v[a] * 0.98f;
    if (m->color_spheres_mode == 1u && !hide) {
        float e = eased_boost(m->color_spheres_boost_time, m->color_spheres_boost_duration);
        float damping = 0.98f * (1.0f - e) + 0.7f * e;
        for (int a = 0; a < 3; a++) v[a] = v[a] * damping;
    }
    for (int k = 0; k < ORC_SEGMENTS; k++) {
        if (m->segment_mode[k] != 1u || hide) continue;
        float e = eased_boost(m->segment_boost_time[k], m->segment_boost_duration[k]);
        float damping = 0.98f * (1.0f - e) + 0.7f * e;
        for (int a = 0; a < 3; a++) v[a] = v[a] * damping;
    }

    /* SIM:1479-1513 hidden particles are parked at the x faces.  `gridSize.x.mul(0.1)` has an
     * int left operand, and three.js types a binary operator by its left operand unless the
     * right one is a longer vector, building both operands in that type: the literal becomes
     * i32(0.1) = 0, the "edge region" has width 0, and the targets are the planes x = 0 and
     * x = gridSize.x.  The later `.div(2.0)` on these ints is an integer division by 2. */
    if (hide) {
        const int edge_region_width = u->grid_x * (int)0.1f;
        const int left_edge_max = edge_region_width;
        const int right_edge_min = u->grid_x - edge_region_width;
        const float hash = (float)(uint32_t)index * 12.9898f;
        const int use_left_edge = fractf(hash) < 0.5f;
        const int target_x = use_left_edge ? left_edge_max / 2 : right_edge_min + edge_region_width / 2;
        const float dist = fabsf(pos[0] - (float)target_x);
        if (dist > (float)(edge_region_width / 2)) {
            const float toward = pos[0] < (float)target_x ? 1.0f : -1.0f;
            const float excess = dist - (float)(edge_region_width / 2);
            const float strength = excess * 0.3f;
            v[0] = v[0] + (toward * strength) * dt;
        }
    }

    /* SIM:1515-1533 */
    for (int a = 0; a < 3; a++) {
        pos[a] = pos[a] + v[a] * dt;
        pos[a] = fminf(fmaxf(pos[a], 2.0f), (float)(gsi[a] - 2));
    }
    for (int a = 0; a < 3; a++) {
        float xn = pos[a] + (v[a] * dt) * 3.0f;
        float wall_min = 3.0f, wall_max = gs[a] - 3.0f;
        if (xn < wall_min) v[a] = v[a] + (wall_min - xn) * 0.3f;
        if (xn > wall_max) v[a] = v[a] + (wall_max - xn) * 0.3f;
    }
    for (int a = 0; a < 3; a++) {
        P[ORC_POS + a] = pos[a];
        P[ORC_VEL + a] = v[a];
        P[ORC_DIR + a] = P[ORC_DIR + a] * (1.0f - 0.1f) + v[a] * 0.1f;
    }
}

void orc_g2p_modes_ids(float *particles, const int32_t *ids, int64_t count, const float *cellf,
                       const orc_uniforms *u, const orc_mode_uniforms *m)
{
    static const orc_mode_uniforms zero_modes;
    if (!m) m = &zero_modes;
    int64_t n = count >= 0 ? count : (int64_t)u->num_particles;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++)
        g2p_modes_one(particles + i * ORC_STRIDE, ids ? ids[i] : (int32_t)i, cellf, u, m);
}

void orc_g2p_modes(float *particles, const float *cellf, const orc_uniforms *u, const orc_mode_uniforms *m)
{
    orc_g2p_modes_ids(particles, NULL, -1, cellf, u, m);
}

void orc_step_modes(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u,
                    const orc_mode_uniforms *m)
{
    int64_t count = (int64_t)u->grid_x * u->grid_y * u->grid_z;
    orc_clear_grid(cell, cellf, count);
    orc_p2g1(particles, cell, u);
    orc_p2g2(particles, cell, u);
    orc_update_grid(cell, cellf, u);
    orc_g2p_modes(particles, cellf, u, m);
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
