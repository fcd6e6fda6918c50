/*
 * CPU oracle for the MLS-MPM step of noahschwad/flow -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of the reference's five compute kernels
 * (src/mls-mpm/mlsMpmSimulator.js:262-429, 431-443, 484-494, 1316-1362, 1515-1533)
 * and of triNoise3Dvec (src/common/noise.js:3-80), over the reference's own AoS
 * particle record (src/mls-mpm/structuredArray.js:83-119, 28 floats) and its two
 * grids (mlsMpmSimulator.js:131-139).  Nothing in the product path may link,
 * import or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures, golden vectors or
 * recorded outputs for this path, and its execution needs three@0.176.0 plus a
 * WebGPU implementation, neither of which exists here.  The oracle is therefore
 * pinned only by (a) an independent numpy restatement (oracle/mpm_numpy.py) that
 * must agree bit for bit, and (b) implementation-independent properties of the
 * reference's arithmetic (see tests/test_oracle.py).
 *
 * The mode branches of g2p (orc_g2p_modes) have their numpy twin too (mpm_numpy.g2p_modes).
 * Two of their expressions depend on how three.js types a binary operator whose operands differ
 * (`gridSize.sub(1)` at SIM:490 and `gridSize.x.mul(0.1)` at SIM:1481): this restatement
 * follows three.js's OperatorNode, which takes the left operand's type unless the right one
 * is a longer vector and converts both operands to it.  For SIM:490 both readings agree; for
 * the hide-percentage edge parking (SIM:1479-1513) the int reading makes the edge regions
 * zero wide.  Without three.js itself that reading cannot be confirmed.
 *
 * Arithmetic conventions (WGSL semantics, which is what three.js emits):
 *   - i32(f32): truncate toward zero, saturating, NaN -> 0
 *   - fract(x) = x - floor(x);  mix(a,b,t) = a*(1-t) + b*t
 *   - mat3 * vec3 = (col0*v.x + col1*v.y) + col2*v.z, no fused multiply-add
 *   - normalize(v) = v / sqrt((x*x + y*y) + z*z), IEEE sqrt and divide
 *   - pow(x, 5.0) is restated as ((x*x)*(x*x))*x and pow(x, 2) as x*x
 *   - i32 atomicAdd wraps modulo 2^32
 * Build with -ffp-contract=off and without fast-math.
 */
#ifndef MPM_ORACLE_H
#define MPM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Float offsets inside one particle record (structuredArray.js:83-119 applied to
 * the struct at mlsMpmSimulator.js:78-86). */
enum {
    ORC_POS = 0, ORC_DENSITY = 3, ORC_VEL = 4, ORC_MASS = 7,
    ORC_C = 8,            /* C[col][row] at ORC_C + 4*col + row */
    ORC_DIR = 20, ORC_COLOR = 24, ORC_STRIDE = 28
};

typedef struct {
    int32_t grid_x, grid_y, grid_z;      /* uniforms.gridSize (mlsMpmSimulator.js:148) */
    uint32_t num_particles;              /* uniforms.numParticles (:151) */
    float dt;                            /* :150, set per frame at :2524-2528 */
    float stiffness, rest_density, dynamic_viscosity, noise; /* :143-146 */
    float time;                          /* three.js TSL `time` node read at :1318 */
    float mouse_ray_direction[3];        /* :256 */
    float mouse_ray_origin[3];           /* :257 */
    float mouse_force[3];                /* :258 */
    uint32_t cursor_interaction;         /* :259 */
    float hide_percentage;               /* :260 (orc_g2p ignores it; orc_g2p_modes honours it) */
    uint32_t grid_mode;                  /* :152 (orc_g2p ignores it; orc_g2p_modes honours it) */
    uint32_t front_gravity_mode;         /* :153 (as grid_mode) */
} orc_uniforms;

/* The mode uniforms read by g2p's art-direction forces (SIM:152-254).  Index k of the
 * segment arrays is the population segment in the reference's order: pink, purple, cyan,
 * teal, dark blue, dark purple, dark red, orange, light green (SIM:761-1313). */
#define ORC_SEGMENTS 9
typedef struct {
    uint32_t cylinder_mode, sphere_containment_mode, polygon_containment_mode;
    uint32_t color_spheres_mode, two_color_sphere_mode;
    uint32_t segment_mode[ORC_SEGMENTS];
    float cylinder_radius, cylinder_center_x, cylinder_center_y;          /* SIM:155-157 */
    float sphere_radius, sphere_center[3];                                /* SIM:233, 236-238 */
    uint32_t polygon_vertex_count;                                        /* SIM:172 */
    float polygon_vertices[16][2];                                        /* SIM:173-176 */
    float color_spheres_boost_time, color_spheres_boost_duration;         /* SIM:177-178 */
    float color_sphere_radius;                                            /* SIM:235 */
    float color_sphere_positions[13][3], palette_colors[13][3];           /* SIM:240-245 */
    float segment_boost_time[ORC_SEGMENTS], segment_boost_duration[ORC_SEGMENTS];
    float segment_main_radius[ORC_SEGMENTS], segment_sphere_radius[ORC_SEGMENTS];
    float segment_sphere_center[ORC_SEGMENTS][3], segment_negative_center[ORC_SEGMENTS][3];
    float negative_space_sphere_radius;                                   /* SIM:215 */
    uint32_t grid_width, grid_height;                                     /* SIM:246-247 */
    float grid_start_x, grid_start_y, grid_spacing_x, grid_spacing_y;     /* SIM:248-251 */
    float grid_z, grid_random_offset, grid_noise_strength;                /* SIM:252-254 */
} orc_mode_uniforms;

/* threads <= 1: plain scalar loops.  threads > 1: OpenMP, one iteration per
 * particle / cell as the reference dispatches them, relaxed atomic adds. */
void orc_set_threads(int threads);
int  orc_get_threads(void);

void orc_clear_grid(int32_t *cell, float *cellf, int64_t cell_count);
void orc_p2g1(const float *particles, int32_t *cell, const orc_uniforms *u);
void orc_p2g2(float *particles, int32_t *cell, const orc_uniforms *u);
void orc_update_grid(const int32_t *cell, float *cellf, const orc_uniforms *u);
void orc_g2p(float *particles, const float *cellf, const orc_uniforms *u);
/* g2p with every mode branch of SIM:431-1536, the hide-percentage edge parking included;
 * u->grid_mode, u->front_gravity_mode and u->hide_percentage are honoured (u->num_particles
 * is uniforms.numParticles).  m == NULL means all mode uniforms are zero. */
void orc_g2p_modes(float *particles, const float *cellf, const orc_uniforms *u,
                   const orc_mode_uniforms *m);
/* The same over a subset of the reference's buffer: `count` records, ids[i] being the
 * instanceIndex of record i (a slab of a decomposed run holds some of the particles, in arbitrary
 * order), while u->num_particles stays uniforms.numParticles of the whole buffer.  ids == NULL:
 * i itself; count < 0: u->num_particles records. */
void orc_g2p_modes_ids(float *particles, const int32_t *ids, int64_t count, const float *cellf,
                       const orc_uniforms *u, const orc_mode_uniforms *m);
void orc_step_modes(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u,
                    const orc_mode_uniforms *m);

/* One reference update(): the five kernels in submission order
 * (mlsMpmSimulator.js:2538-2539). */
void orc_step(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u);

/* n_steps updates; the time uniform of step k is (float)(u->time + k*dtime). */
void orc_run(float *particles, int32_t *cell, float *cellf, const orc_uniforms *u,
             int n_steps, double dtime);

void orc_tri_noise3d(const float pos[3], float speed, float time, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
