/*
 * flowmpm -- C ABI of the MI355X-native MLS-MPM step for noahschwad/flow.
 *
 * The reference has no FFI: its simulator is a JavaScript class whose five
 * compute kernels are handed to three.js (src/mls-mpm/mlsMpmSimulator.js:2538-2539,
 * `renderer.computeAsync([clearGrid, p2g1, p2g2, updateGrid, g2p])`).  This header
 * is the boundary a host binding (Node-API addon, ctypes, cgo ...) attaches to in
 * place of that call.  Each entry point cites the reference code it replaces;
 * paths are relative to the reference tree and SIM abbreviates
 * src/mls-mpm/mlsMpmSimulator.js.
 *
 * Conventions: every function returns an int status (FLOWMPM_OK = 0); no C++
 * exception crosses the boundary; `flowmpm_last_error` returns the message of the
 * most recent failure on a handle.  Host arrays are caller-owned and copied.  A
 * handle is not thread-safe.  All work of a handle is enqueued on one HIP stream.
 */
#ifndef FLOWMPM_H
#define FLOWMPM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOWMPM_OK               0
#define FLOWMPM_ERR_INVALID      1   /* bad argument or call order */
#define FLOWMPM_ERR_HIP          2   /* HIP runtime failure (message has the HIP error) */
#define FLOWMPM_ERR_UNSUPPORTED  3   /* a reference mode this library does not implement */
#define FLOWMPM_ERR_CAPACITY     4   /* more particles than the handle was created for */

/* Particle record the host exchanges with the library: the reference's AoS layout,
 * 28 floats (src/mls-mpm/structuredArray.js:83-119 applied to SIM:78-86).
 * Float offsets: position 0-2, density 3, velocity 4-6, mass 7,
 * C[col][row] at 8 + 4*col + row, direction 20-22, color 24-26; 23 and 27 are pad. */
#define FLOWMPM_PARTICLE_STRIDE 28

/* Kernel ids for flowmpm_run_kernel, in the reference's submission order (SIM:2538). */
#define FLOWMPM_KERNEL_CLEAR_GRID   0   /* SIM:262-277  */
#define FLOWMPM_KERNEL_P2G1         1   /* SIM:295-333  */
#define FLOWMPM_KERNEL_P2G2         2   /* SIM:336-392  */
#define FLOWMPM_KERNEL_UPDATE_GRID  3   /* SIM:395-429  */
#define FLOWMPM_KERNEL_G2P          4   /* SIM:431-1536 (chaos path) */
#define FLOWMPM_KERNEL_COUNT        5
/* The tiled algorithm's two-pass form of p2g1 + p2g2 (no reference counterpart one by one:
 * MASS adds only the mass words of SIM:329; MOMENTUM adds, per particle and cell, the sum of
 * the separately encoded momentum words of SIM:326-328 and SIM:386-388).  After both, the
 * grid equals the reference's grid after p2g2 bit for bit.  flowmpm_step uses this form. */
#define FLOWMPM_KERNEL_P2G_MASS     16
#define FLOWMPM_KERNEL_P2G_MOMENTUM 17
/* clearGrid of the fixed-point grid only (20 instead of 36 bytes per cell).  The float grid is
 * then kept by the next FLOWMPM_KERNEL_UPDATE_GRID, which also writes the cleared value into
 * the cells that held mass at the previous updateGrid and hold none now; after it the float
 * grid equals the reference's bit for bit.  Timed as FLOWMPM_KERNEL_CLEAR_GRID.  flowmpm_step
 * uses this form. */
#define FLOWMPM_KERNEL_CLEAR_FIXED_GRID 18
/* Slab handles of the tiled algorithm: the mass pass split around particle migration, so that it
 * can be enqueued before the host knows the migration counts.  _STAYERS is FLOWMPM_KERNEL_P2G_MASS
 * over the resident particles except those whose floor(x) has left the slab's owned planes (the
 * leavers flowmpm_pack_leavers will remove); _ARRIVALS adds the particles inserted by
 * flowmpm_insert_arrivals since then.  Together they leave the grid FLOWMPM_KERNEL_P2G_MASS would
 * leave after the migration.  Both are timed as FLOWMPM_KERNEL_P2G1. */
#define FLOWMPM_KERNEL_P2G_MASS_STAYERS  19
#define FLOWMPM_KERNEL_P2G_MASS_ARRIVALS 20
/* Slab handles of the tiled algorithm: the momentum pass and g2p in two launches, so that a halo
 * exchange can be in flight during the first.  _INTERIOR runs the work items whose tile
 * neighbourhood cannot reach the four planes a halo exchange changes (slab_x0 - 1, slab_x0,
 * slab_x0 + slab_nx - 1, slab_x0 + slab_nx); _BOUNDARY runs the others and everything the tiled
 * kernels leave to the follow-up kernel.  _INTERIOR then _BOUNDARY equals the whole stage.
 * FLOWMPM_KERNEL_UPDATE_GRID_BOUNDARY is updateGrid over the tiles that hold those four planes
 * again (updateGrid is a pure function of the fixed-point grid, so running it early over
 * everything and once more over these tiles after the momentum halo has been added gives the
 * float grid of a single late run). */
#define FLOWMPM_KERNEL_P2G_MOMENTUM_INTERIOR 21
#define FLOWMPM_KERNEL_P2G_MOMENTUM_BOUNDARY 22
#define FLOWMPM_KERNEL_G2P_INTERIOR          23
#define FLOWMPM_KERNEL_G2P_BOUNDARY          24
#define FLOWMPM_KERNEL_UPDATE_GRID_BOUNDARY  25

/* Timing slots reported by flowmpm_get_timings: the five kernels, then the
 * particle re-binning pass of the tiled algorithm (no reference counterpart). */
#define FLOWMPM_TIMING_REBIN        5   /* the start-of-step re-binning sequence (runs every step; most return at once) */
#define FLOWMPM_TIMING_REBINS_DONE  6   /* the re-bins the device actually performed: launches[] their number,
                                          ms[] the device time they took (device wall clock, every step,
                                          whatever the profiling stride) */
#define FLOWMPM_TIMING_STRAGGLERS   7   /* launches[]: particles the tiled g2p stages found outside their tile's
                                          arena (or past the binned range) and handed to the follow-up kernels,
                                          summed over the steps; ms[] unused */
#define FLOWMPM_TIMING_SLOTS        8

/* How the particle-to-grid scatter is organised. */
#define FLOWMPM_ALGO_SCATTER 0   /* the reference's algorithm: one thread per particle,
                                    27 x 4 global i32 atomics (SIM:315-331) */
#define FLOWMPM_ALGO_TILED   1   /* particles binned by grid tile, contributions summed
                                    as fixed-point integers in LDS, one flush per tile;
                                    flowmpm_step runs p2g as the MASS + MOMENTUM passes */
#define FLOWMPM_ALGO_TILED_UNFUSED 2  /* same, but flowmpm_step keeps p2g1 and p2g2 as the
                                    reference stages them (for A/B measurements) */

/* Per-particle fields for flowmpm_download_field (what the reference's renderers
 * read from the particle buffer: src/mls-mpm/particleRenderer.js:119-133,
 * src/mls-mpm/pointRenderer.js:17-19, plus velocity/direction). */
#define FLOWMPM_FIELD_POSITION  0   /* 3 floats */
#define FLOWMPM_FIELD_VELOCITY  1   /* 3 floats */
#define FLOWMPM_FIELD_DENSITY   2   /* 1 float  */
#define FLOWMPM_FIELD_COLOR     3   /* 3 floats */
#define FLOWMPM_FIELD_DIRECTION 4   /* 3 floats */
#define FLOWMPM_FIELD_MASS      5   /* 1 float  */

/* Boundary planes of a slab's local grid for the halo exchange. */
#define FLOWMPM_PLANE_LOW_HALO    0  /* global x = slab_x0 - 1          */
#define FLOWMPM_PLANE_LOW_OWNED   1  /* global x = slab_x0              */
#define FLOWMPM_PLANE_HIGH_OWNED  2  /* global x = slab_x0 + slab_nx - 1 */
#define FLOWMPM_PLANE_HIGH_HALO   3  /* global x = slab_x0 + slab_nx     */

/* Sides of a slab for the packed halo exchange. */
#define FLOWMPM_SIDE_LOW   0   /* planes slab_x0 - 1 and slab_x0                       */
#define FLOWMPM_SIDE_HIGH  1   /* planes slab_x0 + slab_nx - 1 and slab_x0 + slab_nx   */

/* Component masks for flowmpm_plane_add_i32 and the flowmpm_halo_* calls. */
#define FLOWMPM_COMP_MOMENTUM 0x7
#define FLOWMPM_COMP_MASS     0x8
#define FLOWMPM_COMP_ALL      0xF

typedef struct flowmpm_sim flowmpm_sim;

/* Replaces the allocations of init() (SIM:74-139): gridSize, the particle buffer of
 * `max_particles` records and the two grids. */
typedef struct {
    int32_t grid_x, grid_y, grid_z;   /* uniforms.gridSize, SIM:76, 148 (global grid) */
    int32_t slab_x0, slab_nx;         /* x-planes this handle owns; slab_nx = 0 means the
                                         whole grid (single GPU) */
    int32_t device;                   /* HIP device ordinal */
    int32_t algorithm;                /* FLOWMPM_ALGO_* */
    int32_t rebin_interval;           /* tiled: re-bin particles at least every k steps (>= 1); the
                                       * device re-bins earlier when too many particles have left
                                       * their tile's neighbourhood.  Never affects results. */
    int64_t max_particles;            /* conf.maxParticles, src/conf.js:8; at most 2^30 - 64 */
    void   *stream;                   /* hipStream_t to enqueue on when external_stream != 0
                                         (NULL is then the device's default stream) */
    int32_t external_stream;          /* 0: the handle creates and owns its stream */
    int32_t slab_max_nx;              /* slab handles: the most planes flowmpm_set_slab may give this
                                         handle later (its grids are allocated for that many); 0 means
                                         slab_nx, i.e. a slab that is never re-cut */
} flowmpm_config;

/* The uniforms the hot path reads (SIM:141-151, 256-260) plus the clock that
 * three.js supplies through its TSL `time` node (SIM:22, 1318). */
typedef struct {
    float dt;                         /* SIM:150, 2524-2528 */
    float stiffness;                  /* SIM:143 */
    float rest_density;               /* SIM:144 */
    float dynamic_viscosity;          /* SIM:145 */
    float noise;                      /* SIM:146 */
    float time;                       /* TSL time node */
    float mouse_ray_direction[3];     /* SIM:256, 1543 */
    float mouse_ray_origin[3];        /* SIM:257, 1544 */
    float mouse_force[3];             /* SIM:258, 2533 */
    uint32_t cursor_interaction;      /* SIM:259 */
    float hide_percentage;            /* SIM:260, read at SIM:441-442; slab handles take 0 only
                                       * (FLOWMPM_ERR_UNSUPPORTED otherwise) until
                                       * flowmpm_set_domain_particles has been called */
    uint32_t grid_mode;               /* SIM:152, read at SIM:444 */
    uint32_t front_gravity_mode;      /* SIM:153, read at SIM:485, 1316, 1365 */
} flowmpm_uniforms;

/* The mode uniforms g2p reads for its containment / attraction forces (SIM:154-254).
 * Entry k of the segment arrays is the k-th population-segment mode in the order of
 * SIM:761-1313: pink, purple, cyan, teal, dark blue, dark purple, dark red, orange,
 * light green (uniforms populationSegment<Name>Mode, ...BoostTime, ...BoostDuration,
 * populationMainSphereRadius<Name>, <name>SphereCenterX/Y/Z, <name>SphereRadius,
 * <name>NegativeSpaceCenterX/Y/Z).  All zero after flowmpm_create. */
#define FLOWMPM_SEGMENTS 9
typedef struct {
    uint32_t cylinder_mode;                 /* SIM:154 */
    uint32_t sphere_containment_mode;       /* SIM:158 */
    uint32_t polygon_containment_mode;      /* SIM:159 */
    uint32_t color_spheres_mode;            /* SIM:160 */
    uint32_t two_color_sphere_mode;         /* SIM:161 */
    uint32_t segment_mode[FLOWMPM_SEGMENTS];              /* SIM:162-170 */
    float cylinder_radius, cylinder_center_x, cylinder_center_y;   /* SIM:155-157 */
    float sphere_radius;                    /* SIM:233 */
    float sphere_center[3];                 /* SIM:236-238 */
    uint32_t polygon_vertex_count;          /* SIM:172; at most 16 */
    float polygon_vertices[16][2];          /* SIM:173-176 */
    float color_spheres_boost_time;         /* SIM:177 */
    float color_spheres_boost_duration;     /* SIM:178 */
    float color_sphere_radius;              /* SIM:235 */
    float color_sphere_positions[13][3];    /* SIM:240-242 */
    float palette_colors[13][3];            /* SIM:243-245 */
    float segment_boost_time[FLOWMPM_SEGMENTS];           /* SIM:179-195, odd lines */
    float segment_boost_duration[FLOWMPM_SEGMENTS];       /* SIM:180-196, even lines */
    float segment_main_radius[FLOWMPM_SEGMENTS];          /* SIM:225-232, 234 */
    float segment_sphere_radius[FLOWMPM_SEGMENTS];        /* SIM:197-213, odd lines */
    float segment_sphere_center[FLOWMPM_SEGMENTS][3];     /* SIM:198-214, even lines */
    float segment_negative_center[FLOWMPM_SEGMENTS][3];   /* SIM:216-224 */
    float negative_space_sphere_radius;     /* SIM:215 */
    uint32_t grid_width, grid_height;       /* SIM:246-247 */
    float grid_start_x, grid_start_y;       /* SIM:248-249 */
    float grid_spacing_x, grid_spacing_y;   /* SIM:250-251 */
    float grid_z;                           /* SIM:252 */
    float grid_random_offset;               /* SIM:253 */
    float grid_noise_strength;              /* SIM:254 */
} flowmpm_mode_uniforms;

typedef struct {
    double  ms[FLOWMPM_TIMING_SLOTS];        /* summed HIP-event time per slot */
    int64_t launches[FLOWMPM_TIMING_SLOTS];  /* how many timed launches per slot */
} flowmpm_timings;

/* Library version and the number of visible HIP devices (fails when no GPU). */
int flowmpm_version(void);
int flowmpm_device_count(int *count);

/* init(): SIM:74-139. */
int flowmpm_create(const flowmpm_config *config, flowmpm_sim **out);
int flowmpm_destroy(flowmpm_sim *sim);
const char *flowmpm_last_error(const flowmpm_sim *sim);   /* sim may be NULL: last
                                                             create/global failure */

/* Replaces three.js's first upload of particleBuffer.floatArray (SIM:87, 126-128) and
 * the forced re-upload after a mode switch (SIM:2340-2366).  `aos` holds `count`
 * records; record i becomes particle id i.  In a slab handle only the caller's
 * particles for that slab are passed, with their global ids in `ids` (NULL: 0..count-1). */
int flowmpm_upload_particles(flowmpm_sim *sim, const float *aos, const int32_t *ids,
                             int64_t count);
/* The host loop of init() that fills the particle buffer (SIM:114-129), run on the device with the
 * seeded counter-based generator of the hosts (flow_amd/particles.py, host_js/particles.js) in place of
 * Math.random: the particles with ids [first_id, first_id + count) of scene `scene`, bit-identical to the
 * hosts' records, replace the handle's particles.  For domains whose records would not fit host memory
 * comfortably (2^27 particles are 15 GB of AoS records).  A single-domain handle takes first_id = 0 and
 * stores particle i in slot i; a slab handle keeps the particles whose floor(x) lies in its owned planes,
 * in no particular order (results never depend on the order).  *stored (may be NULL) receives the
 * number kept.  FLOWMPM_SCENE_DAM_BREAK (no reference counterpart; benchmark configuration 5) spreads
 * the particles uniformly over x in [3, fill * grid_x], y in [3, grid_y - 3], z in [3, grid_z - 3];
 * `fill` is ignored by the ellipsoid. */
#define FLOWMPM_SCENE_ELLIPSOID 0   /* SIM:114-129 */
#define FLOWMPM_SCENE_DAM_BREAK 1
int flowmpm_generate_particles(flowmpm_sim *sim, int32_t scene, uint32_t seed, int64_t first_id,
                               int64_t count, double fill, int64_t *stored);
/* uniforms.numParticles / dispatch counts (SIM:2513-2522): simulate the first
 * `count` particle ids only.  Single-GPU handles only. */
int flowmpm_set_num_particles(flowmpm_sim *sim, int64_t count);
int flowmpm_get_num_particles(flowmpm_sim *sim, int64_t *live, int64_t *resident);
/* Slab handles only: uniforms.numParticles of the whole domain (SIM:151, 2513-2522), which the
 * hide-percentage branch of g2p divides the particle id by (SIM:444-483).  A slab only knows its
 * resident particles, so hidePercentage > 0 is accepted on a slab handle once this is set. */
int flowmpm_set_domain_particles(flowmpm_sim *sim, int64_t count);

/* The uniform writes of update() (SIM:2496-2534). */
int flowmpm_set_uniforms(flowmpm_sim *sim, const flowmpm_uniforms *u);

/* The uniform writes of toggleGridMode and its set-up helpers (SIM:1679-2493) and the
 * per-frame boost clocks (SIM:2500-2521).  While every mode flag here and gridMode /
 * frontGravityMode are 0, g2p runs its plain path and these values are never read.
 * polygon_vertex_count > 16 is FLOWMPM_ERR_INVALID. */
int flowmpm_set_mode_uniforms(flowmpm_sim *sim, const flowmpm_mode_uniforms *m);

/* Tuning of the tiled algorithms (no reference counterpart; results never depend on any of them).
 * RIDE_REBINS: how flowmpm_step of a single-domain FLOWMPM_ALGO_TILED handle re-bins -- 2 (default) the
 *   re-bin rides on the step and the step's mass pass counts the tiles, 1 it rides with a counting
 *   launch of its own, 0 the standalone two-launch re-bin (what flowmpm_run_kernel and slab handles
 *   always use).  The flowmpm_slab_step_* sequence of a slab handle rides in form 2 (or 0).
 * DRIFT_DIVISOR: the device re-bins when the particles outside their tile's arena, summed over the steps
 *   since the last re-bin, exceed particles / divisor (default 4096 riding, 256 riding on a slab, 256
 *   standalone, 48 standalone on a slab;
 *   0 leaves only rebin_interval).
 * CHUNK: most particles per work item (256 .. 4096, default 4096).  SPARSE_TILE: tiles with fewer
 *   particles than this get no work item and go down the follow-up kernels' path (default 0).  Both
 *   force a re-bin at the next step. */
#define FLOWMPM_OPTION_RIDE_REBINS   0
#define FLOWMPM_OPTION_DRIFT_DIVISOR 1
#define FLOWMPM_OPTION_CHUNK         2
#define FLOWMPM_OPTION_SPARSE_TILE   3
int flowmpm_set_option(flowmpm_sim *sim, int32_t option, int64_t value);

/* renderer.computeAsync([...5 kernels]) (SIM:2538-2539), `n_steps` times.  The clock
 * of step k is (float)(uniforms.time + k * dtime).  Asynchronous: returns after
 * enqueueing; flowmpm_synchronize or any download waits. */
int flowmpm_step(flowmpm_sim *sim, int32_t n_steps, double dtime);
/* One of the five kernels alone, for parity tests against the reference's kernels. */
int flowmpm_run_kernel(flowmpm_sim *sim, int32_t kernel_id);
int flowmpm_synchronize(flowmpm_sim *sim);

/* Read-back.  The reference never reads back; its renderers bind the storage buffer
 * (src/mls-mpm/particleRenderer.js:126-133).  A host renderer or a parity dump uses
 * these.  Particles come back in particle-id order.  A slab handle returns its
 * resident particles in unspecified order together with their global ids. */
int flowmpm_download_particles(flowmpm_sim *sim, float *aos, int32_t *ids, int64_t capacity,
                               int64_t *count);
int flowmpm_download_field(flowmpm_sim *sim, int32_t field, float *out, int64_t capacity);
/* cellBuffer (SIM:131-138): 4 x i32 per cell {x, y, z, mass}; cellBufferF (SIM:139):
 * vec4 {vx, vy, vz, mass}.  Local planes of the handle, x-major (SIM:286-290). */
int flowmpm_get_local_grid(flowmpm_sim *sim, int32_t *x_first, int32_t *x_planes);
int flowmpm_download_grid_i32(flowmpm_sim *sim, int32_t *out, int64_t capacity_cells);
int flowmpm_download_grid_f32(flowmpm_sim *sim, float *out, int64_t capacity_cells);

/* Conservation tracking: the 64-bit sums of the fixed-point grid's x, y, z and mass words over
 * the cells this handle owns (halo planes excluded), computed on the device.  After p2g the
 * mass sum is an integer the rest of the step conserves exactly; summed over the slabs of a
 * decomposed run it equals the single-GPU value. */
int flowmpm_grid_sums(flowmpm_sim *sim, int64_t sums[4]);

/* The same check from the particles' side: the sum over the resident particles and their 27 stencil
 * cells of encodeFixedPoint(weight) (SIM:279-281, 329), in 64 bits, computed from the positions alone.
 * Taken before a step it equals the mass word of flowmpm_grid_sums after that step's p2g (summed over
 * the slabs of a decomposed run). */
int flowmpm_stencil_mass_sum(flowmpm_sim *sim, int64_t *sum);
/* Resident particles per x-plane of the global grid: out[ix] = particles with floor(position.x) = ix,
 * `planes` = grid_x entries (at most 8192).  What a host cuts balanced slabs from (summed over the
 * slabs of a decomposed run). */
int flowmpm_plane_histogram(flowmpm_sim *sim, int64_t *out, int32_t planes);

/* The same histogram for particles [first_id, first_id + count) of a generated scene
 * (flowmpm_generate_particles) in a grid of grid[0] x grid[1] x grid[2] cells, computed on `device`
 * straight from the generator: no handle and no particle store is needed, so a host can choose the
 * slab boundaries before it creates the slab handles. */
int flowmpm_scene_plane_histogram(int32_t device, int32_t scene, uint32_t seed, int64_t first_id,
                                  int64_t count, const int32_t grid[3], double fill, int64_t *out);

/* sizeof of the public structs as this library was compiled: flowmpm_config, flowmpm_uniforms,
 * flowmpm_mode_uniforms, flowmpm_timings.  A binding that re-declares them (ctypes, a typed-array
 * packer) checks its own layout against these at load time. */
int flowmpm_abi_sizes(int64_t sizes[4]);

/* Per-kernel device timing: HIP events attached to the kernel dispatches on the handle's
 * stream (start and end timestamps of the kernel itself).  every_nth_step = 0 switches it
 * off, 1 times every launch, k > 1 times the launches of every k-th step of flowmpm_step
 * (a timed dispatch costs the queue a few microseconds; sampling keeps a timed run within a
 * fraction of a percent of an untimed one).  For flowmpm_run_kernel a step begins at each
 * clearGrid (either form): the launches from the k-th such call up to the next one are
 * timed. */
int flowmpm_set_profiling(flowmpm_sim *sim, int32_t every_nth_step);
int flowmpm_get_timings(flowmpm_sim *sim, flowmpm_timings *out);
int flowmpm_reset_timings(flowmpm_sim *sim);

/* The read-backs above replace the per-frame buffer binding of the reference's renderers
 * (src/mls-mpm/particleRenderer.js:126-133, src/mls-mpm/pointRenderer.js:17-19), so they are called
 * every frame: each handle stages them through one device buffer and one pinned host buffer that
 * grow geometrically and live until flowmpm_destroy.  `count` receives the number of device / pinned
 * allocations the handle has made for staging (and for the leaver lists of a slab) since
 * flowmpm_create; it stops growing once the buffers have reached the size of the largest transfer. */
int flowmpm_get_allocation_count(flowmpm_sim *sim, int64_t *count);

/* flowmpm_download_field for callers that cannot allocate (tight per-frame loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Same arguments and results as flowmpm_download_field,
 * but the id-ordered values are staged in a static device array of the library (3 MiB per GPU the
 * library has created a handle on, held until the process ends) instead of the handle's staging buffer,
 * and nothing is allocated or freed: `out` may be device memory, pinned host memory, or pageable host
 * memory (which the HIP runtime stages through its own pool).  The call waits for its copy (it
 * synchronises the handle's stream), so it cannot be recorded into a HIP graph.  At most 262,144 live
 * particles (conf.maxParticles, src/conf.js:8); more is FLOWMPM_ERR_CAPACITY.  All handles on one GPU
 * share that GPU's array: calls that overlap in time on one GPU race; calls from one host thread never
 * overlap, because each waits for its copy. */
int flowmpm_download_field_static(flowmpm_sim *sim, int32_t field, float *out, int64_t capacity);

/* ---- Slab decomposition (no reference counterpart: the reference is single-GPU).
 * The caller owns the transport (RCCL through torch.distributed) and its device buffers;
 * these are the device-side pieces. ---- */

/* Byte size of one boundary plane as exchanged: grid_y * grid_z records {x, y, z, mass}. */
int flowmpm_plane_bytes(flowmpm_sim *sim, int64_t *bytes);
/* grid[plane].comp += src.comp for the components in `comp_mask`, as wrapping i32 adds;
 * `dev_src` is a device buffer holding one plane of {x, y, z, mass} records (SIM:131-138). */
int flowmpm_plane_add_i32(flowmpm_sim *sim, int32_t plane, const void *dev_src,
                          uint32_t comp_mask);
/* Writes one boundary plane as {x, y, z, mass} records into a caller's device buffer (send
 * staging). */
int flowmpm_plane_copy_out(flowmpm_sim *sim, int32_t plane, void *dev_dst);

/* The halo exchange as the step uses it: the two boundary planes of one side (adjacent in
 * memory, in increasing x; a slab's HIGH pair is its high neighbour's LOW pair) travel with
 * only the components the merge needs, k = popcount(comp_mask) i32 words per cell in
 * x, y, z, mass order: 4 bytes per cell after the mass pass, 12 after the momentum pass,
 * instead of two 16-byte planes each time.  flowmpm_halo_words gives the buffer size in
 * words; pack writes a send buffer, add merges a received one (wrapping i32 adds). */
int flowmpm_halo_words(flowmpm_sim *sim, uint32_t comp_mask, int64_t *words);
int flowmpm_halo_pack(flowmpm_sim *sim, int32_t side, uint32_t comp_mask, void *dev_dst);
int flowmpm_halo_add(flowmpm_sim *sim, int32_t side, uint32_t comp_mask, const void *dev_src);
/* Splits the resident particles after g2p: those whose floor(position.x) left
 * [slab_x0, slab_x0 + slab_nx) are removed from the handle and written as packed
 * records (FLOWMPM_MIGRATE_WORDS 32-bit words each: the 23 fields, then the id) into
 * `dev_out`, grouped by destination rank under the equal-width split given by
 * `plane_starts[0..n_ranks]`.  counts[r] receives the number bound for rank r. */
#define FLOWMPM_MIGRATE_WORDS 24
int flowmpm_extract_leavers(flowmpm_sim *sim, const int32_t *plane_starts, int32_t n_ranks,
                            void *dev_out, int64_t capacity_records, int64_t *counts);
/* The same in two halves, so that the caller can take the counts through its own count
 * exchange and keep the stream busy meanwhile.  flowmpm_find_leavers only enqueues the
 * classification; the n_ranks leaver counts appear as int64 words in `dev_counts` (device
 * memory of the caller) in stream order.  flowmpm_pack_leavers takes those counts back from
 * the caller (host memory) once it has read them, packs the records and compacts the store.
 * Between the two calls only clearGrid and FLOWMPM_KERNEL_P2G_MASS_STAYERS may run; as that stage
 * may re-bin (move) the particles, flowmpm_find_leavers is then called once more before the pack
 * with dev_counts NULL (the counts are the same): that call repeats the classification only if a
 * re-bin has happened since the first one, which the device decides. */
int flowmpm_find_leavers(flowmpm_sim *sim, const int32_t *plane_starts, int32_t n_ranks, void *dev_counts);
int flowmpm_pack_leavers(flowmpm_sim *sim, const int64_t *counts, int32_t n_ranks, void *dev_out,
                         int64_t capacity_records);
/* Appends packed records that arrived from other ranks. */
int flowmpm_insert_arrivals(flowmpm_sim *sim, const void *dev_in, int64_t count);

/* Re-cuts a slab handle: from now on it owns planes [slab_x0, slab_x0 + slab_nx) (slab_nx at most the
 * slab_max_nx it was created with).  Called between steps, on every slab of the decomposition with the
 * new cut, and followed by flowmpm_find_leavers with the new plane_starts: the particles of the planes
 * that changed hands are then leavers and travel through the ordinary migration of the next step
 * (flowmpm_pack_leavers / flowmpm_insert_arrivals), so a re-cut costs one step's migration of those
 * planes and a forced re-bin.  Both grids of the handle are cleared (every step rebuilds them).  What a
 * host does every few steps from the summed flowmpm_plane_histogram when the fluid moves through the
 * domain (the dam-break configuration); results do not depend on the cut. */
int flowmpm_set_slab(flowmpm_sim *sim, int32_t slab_x0, int32_t slab_nx);

/* ---- One step of a FLOWMPM_ALGO_TILED slab handle in six calls, in this order (any other order is
 * FLOWMPM_ERR_INVALID).  Each call only enqueues; the gaps between them are where the host's transport
 * acts.  This is the stage sequence above (clearGrid, mass pass of the stayers, ... g2p of the boundary
 * items, classification of the leavers) with two differences: the host makes 6 calls per step instead of
 * about 20, and, because the library owns the order of the stages, a re-bin found due RIDES on the step
 * as it does in flowmpm_step -- the mass pass counts the tiles (arrivals included), the momentum pass
 * and g2p store to the particles' new slots -- instead of moving every record in two launches of its own.
 *
 *   flowmpm_slab_step_begin          clearGrid of the fixed-point grid, mass pass of the particles that stay
 *     [host: reads the migration counts, flowmpm_pack_leavers, sends / receives, flowmpm_insert_arrivals]
 *   flowmpm_slab_step_pack_mass      mass pass of the arrivals; packs the mass halo of each side
 *     [host: starts the exchange of the mass halo]
 *   flowmpm_slab_step_interior       momentum pass of the work items that cannot see a boundary plane
 *     [host: makes the stream wait for the mass halo]
 *   flowmpm_slab_step_boundary       adds the received mass halo, momentum pass of the other items, packs
 *                                    the momentum halo
 *     [host: starts the exchange of the momentum halo]
 *   flowmpm_slab_step_grid           updateGrid, g2p of the interior items
 *     [host: makes the stream wait for the momentum halo]
 *   flowmpm_slab_step_end            adds the received momentum halo, updateGrid over the boundary tiles
 *                                    again, g2p of the other items, then flowmpm_find_leavers(plane_starts,
 *                                    n_ranks, dev_counts) (skipped when plane_starts is NULL)
 *
 * send_* / recv_*: device buffers of flowmpm_halo_words(FLOWMPM_COMP_MASS resp. FLOWMPM_COMP_MOMENTUM)
 * words; NULL where the slab has no neighbour.  flowmpm_pack_leavers repeats the classification by itself
 * when a re-bin has moved the particles since flowmpm_find_leavers. */
int flowmpm_slab_step_begin(flowmpm_sim *sim);
int flowmpm_slab_step_pack_mass(flowmpm_sim *sim, void *send_low, void *send_high);
int flowmpm_slab_step_interior(flowmpm_sim *sim);
int flowmpm_slab_step_boundary(flowmpm_sim *sim, const void *recv_low, const void *recv_high,
                               void *send_low, void *send_high);
int flowmpm_slab_step_grid(flowmpm_sim *sim);
int flowmpm_slab_step_end(flowmpm_sim *sim, const void *recv_low, const void *recv_high,
                          const int32_t *plane_starts, int32_t n_ranks, void *dev_counts);

/* ---- Slab groups: a decomposed domain behind ONE object, driven by one host thread.  The reference's
 * caller is a single-threaded JavaScript program (src/app.js:72-80 creates the simulator, :156 awaits its
 * update once per frame); a host in that language reaches several GPUs through a group: it owns one
 * slab handle per x-slab (plane_starts[0 .. n_slabs], as in flowmpm_extract_leavers), steps them in
 * lock-step with the flowmpm_slab_step_* sequence and moves halo pairs and migrating particles itself.
 * devices[r] (NULL: base->device for all) places slab r.  All slabs on ONE device: one stream, a slab's
 * packed halo pair is merged by its neighbour where it lies, leavers are inserted straight from the
 * sender's buffer.  Every slab on a device of its OWN: RCCL (ncclCommInitAll over the devices, grouped
 * ncclSend / ncclRecv on a communication stream per device, ordered against the compute stream with
 * events, so the exchanges are in flight under the interior work items); librccl.so is loaded when such
 * a group is created.  Any other placement is FLOWMPM_ERR_INVALID.  base supplies grid, algorithm
 * (FLOWMPM_ALGO_TILED) and rebin_interval; max_particles[r] is slab r's capacity.  Results equal the
 * single domain's bit for bit for any number of slabs. ---- */
typedef struct flowmpm_slab_group flowmpm_slab_group;
int flowmpm_slab_group_create(const flowmpm_config *base, const int32_t *plane_starts, int32_t n_slabs,
                              const int32_t *devices, const int64_t *max_particles, flowmpm_slab_group **out);
int flowmpm_slab_group_destroy(flowmpm_slab_group *group);
const char *flowmpm_slab_group_last_error(const flowmpm_slab_group *group);
int flowmpm_slab_group_size(flowmpm_slab_group *group, int32_t *n_slabs, int32_t *uses_rccl);
/* The r-th slab handle, owned by the group (for timings, per-slab read-backs, options). */
int flowmpm_slab_group_slab(flowmpm_slab_group *group, int32_t r, flowmpm_sim **out);
/* flowmpm_upload_particles / flowmpm_generate_particles for the whole domain: every record goes to the
 * slab that owns floor(position.x); record i is particle id i. */
int flowmpm_slab_group_upload(flowmpm_slab_group *group, const float *aos, int64_t count);
int flowmpm_slab_group_generate(flowmpm_slab_group *group, int32_t scene, uint32_t seed, int64_t count, double fill);
int flowmpm_slab_group_set_uniforms(flowmpm_slab_group *group, const flowmpm_uniforms *u);
int flowmpm_slab_group_set_mode_uniforms(flowmpm_slab_group *group, const flowmpm_mode_uniforms *m);
/* flowmpm_step over the decomposed domain (SIM:2538-2539).  Asynchronous except for one wait per step
 * on the migration counts of the step before, which have long arrived when it is reached. */
int flowmpm_slab_group_step(flowmpm_slab_group *group, int32_t n_steps, double dtime);
int flowmpm_slab_group_synchronize(flowmpm_slab_group *group);
int flowmpm_slab_group_get_num_particles(flowmpm_slab_group *group, int64_t *total, int64_t *per_slab);
/* Read-backs in particle-id order, as from a single-domain handle. */
int flowmpm_slab_group_download_particles(flowmpm_slab_group *group, float *aos, int64_t capacity, int64_t *count);
int flowmpm_slab_group_download_field(flowmpm_slab_group *group, int32_t field, float *out, int64_t capacity);
/* flowmpm_grid_sums summed over the slabs, then the particle count: sums[0..2] momentum, [3] mass, [4] particles. */
int flowmpm_slab_group_conservation(flowmpm_slab_group *group, int64_t sums[5]);
/* Particles handed from slab to slab so far, and the farthest destination in slabs (1 = a neighbour). */
int flowmpm_slab_group_migration_stats(flowmpm_slab_group *group, int64_t *migrated, int32_t *max_hop);
/* Re-balances the group: sums flowmpm_plane_histogram over the slabs, cuts the domain at the particle
 * quantiles again, and -- if the cut differs -- re-cuts every slab (flowmpm_set_slab; the group must have
 * been created with base->slab_max_nx large enough) and starts the migration of the planes that changed
 * hands, which the next step completes.  A boundary moves only as far as at most `max_move_fraction` of
 * the mean particle count per slab changes hands through it in one call (0 < fraction <= 1; the leaver
 * buffers hold a quarter of a slab's capacity), so a large imbalance is worked off over several calls.
 * *changed (may be NULL) receives 1 when the cut moved.  Waits for the device (a few times per call):
 * meant for every 10 to 100 steps. */
int flowmpm_slab_group_rebalance(flowmpm_slab_group *group, double max_move_fraction, int32_t *changed);
int flowmpm_slab_group_plane_starts(flowmpm_slab_group *group, int32_t *plane_starts /* n_slabs + 1 */);
/* RCCL in a world of one on `device`: a communicator, `words` 32-bit words sent to itself inside one group
 * call, compared on the host.  What a host runs before it creates a group over several devices. */
int flowmpm_rccl_selftest(int32_t device, int64_t words);

#ifdef __cplusplus
}
#endif
#endif
