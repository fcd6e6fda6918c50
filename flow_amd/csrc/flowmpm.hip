// C ABI of the MI355X MLS-MPM step (declared in include/flowmpm.h).
//
// Host side of the library: owns the device buffers of one handle, validates
// shapes before every launch, and enqueues the kernels of kernels_scatter.h /
// kernels_tiled.h on the handle's HIP stream.  There is no CPU fallback: every
// entry point that computes needs a HIP device and fails with FLOWMPM_ERR_HIP
// otherwise.
#include "../../include/flowmpm.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <mutex>
#include <string>
#include <vector>

#include "kernels_scatter.h"
#include "kernels_tiled.h"
#include "kernels_scene.h"

// ---- staging of flowmpm_download_field_static ------------------------------------------------
// One array per GPU the library runs on, held until the process ends: the id-ordered values of one
// particle field for at most kStaticFieldParticles particles (3 MiB).  In host code `scratch` is only
// a stand-in for the symbol, never a device address: the address on a device comes from
// hipGetSymbolAddress with that device current (scratch_setup, called by flowmpm_create), and host
// code uses that pointer wherever it needs the array's memory.
constexpr long long kStaticFieldParticles = 262144;          // conf.maxParticles, src/conf.js:8
static __device__ __align__(256) float scratch[kStaticFieldParticles * 3];

namespace {

thread_local std::string g_last_error;

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float *g_scratch_ptr[kMaxDevices] = {};       // per device: where that device's copy of `scratch` lives

// With `device` current.  Returns hipSuccess and the device's address of `scratch`.
hipError_t scratch_setup(int device, float **out)
{
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<float *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct EventPair {
    hipEvent_t start, stop;
    int slot;
    bool continuation;      // second launch of one stage: adds its time, not a launch
};

}  // namespace

constexpr int kSlabRideDivisor = 256;      // measured on the 8 slabs of 2^27 particles / 512^3 (tools/divisor_sweep_slab.sh): 2048 -> 2.00, 1024 -> 1.98, 512 -> 1.96, 256 -> 1.91, 128 -> 1.91 ms per slab step

struct flowmpm_sim {
    flowmpm_config cfg{};
    mpm::Params params{};
    flowmpm_uniforms uniforms{};
    flowmpm_mode_uniforms modes{};
    mpm::ModeParams *d_modes = nullptr;   // device copy g2p's mode instantiation reads
    bool modes_stale = true;              // d_modes differs from `modes` / the mode flags of `uniforms`
    long long modes_for_live = -1;        // numParticles held by d_modes
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool slab = false;

    long long capacity = 0;      // slots in the store
    long long resident = 0;      // particles on the device
    long long live = 0;          // simulated particles (== resident except set_num_particles)
    long long domain_particles = 0;   // slab handles: uniforms.numParticles of the whole domain
    long long uploaded = 0;      // records given to upload (single GPU)
    mpm::StoreSel store{};       // tiled: two stores, re-binning gathers from one into the other
    // re-bin when (particles outside their arena) * divisor > particles: drift_divisor for a re-bin that
    // rides on flowmpm_step (cheap, so it starts early), drift_divisor_standalone for the two-launch form
    // (stage-by-stage hosts, slab handles)
    int drift_divisor = 256, drift_divisor_standalone = 256;
    // flowmpm_step of a single-domain handle on the two-pass tiled algorithm lets a re-bin ride on the
    // step's own stores (mpm::RebinCtl::riding) instead of moving every record once more
    // 2: the step's mass pass also counts the tiles; 1: a counting launch of its own; 0: off, the
    // standalone two-launch re-bin (FLOWMPM_RIDE_REBINS sets it, for A/B runs).
    int ride_rebins = 0;
    // The fixed-point grid is double-buffered as two sets {momentum records, mass words}: {cell, mass_i} is
    // the set the current step accumulates into, {cell_spare, mass_spare} the other one.
    int4 *cell = nullptr;        // momentum words (word 3 unused on the device)
    int4 *cell_spare = nullptr;
    int4 *cell_first = nullptr;  // the momentum buffer that owns bit 0 (the other one owns bit 3)
    // {cell_spare, mass_spare} hold nothing but zeros and their tile bits are clear: the updateGrid of
    // flowmpm_step zeroes the set of the step before while it visits its tiles, so the next step needs
    // no clearGrid launch
    bool spare_clean = false;
    int *mass_i = nullptr;       // mass words, one per cell
    // The float grid is non-zero only where this buffer is positive: the mass words of the step
    // whose updateGrid last wrote it.  Either mass_i or mass_spare.
    int *mass_of_cellf = nullptr;
    int *mass_spare = nullptr;
    // Tiled handles: per-tile bits saying which buffer may be non-zero where (mpm::FixedGrid).
    int *dirty = nullptr;
    int *mass_first = nullptr;   // the mass buffer that owns bit 1 (the other one owns bit 2)
    // p2g1 and the mass pass set the bits from the particles they scatter; p2g2 and the momentum
    // pass scatter the same particles over the same cells and rely on that.  False whenever that
    // is not known to hold (after a clear, a g2p, or any change of the particle set).
    bool tiles_marked = false;
    // Slab handles: first slot of the particles inserted since the stayers' mass pass (-1: none),
    // and whether that pass has run on the current particles.
    long long arrivals_first = -1;
    bool stayers_done = false;
    int slab_phase = 0;          // which flowmpm_slab_step_* call comes next (0: _begin)
    // a standalone re-bin may have moved the particles since the leavers were classified (a riding one
    // completes inside its step, before the classification): flowmpm_pack_leavers then lets the device check
    bool moved_since_find = false;
    int mass_bit(const int *buffer) const { return buffer == mass_first ? 2 : 4; }
    int cell_bit(const int4 *buffer) const { return buffer == cell_first ? 1 : 8; }
    mpm::FixedGrid fixed() const
    {
        return mpm::FixedGrid{reinterpret_cast<int *>(cell), mass_i, dirty, cell_bit(cell) | mass_bit(mass_i)};
    }
    // Work items whose arena reaches the planes a halo exchange changes (local planes 0, 1,
    // slab_nx, slab_nx + 1).  The arena of tile column tx spans local planes
    // kTX tx - kRim .. kTX tx + kTX + kRim - 1, so the boundary columns are those with
    // kTX tx - kRim <= 1, i.e. tx <= (1 + kRim) / kTX, and those with kTX tx + kTX + kRim - 1 >= slab_nx,
    // i.e. tx >= ceil((slab_nx - kTX - kRim + 1) / kTX) = (slab_nx - kRim) / kTX (rounded down, not below 0).
    mpm::StagePart stage_part(int part) const
    {
        return mpm::StagePart{part, (1 + mpm::kRim) / mpm::kTX, std::max(cfg.slab_nx - mpm::kRim, 0) / mpm::kTX};
    }
    mpm::TileRange tile_range() const
    {
        return slab ? mpm::TileRange{0, cfg.slab_nx / mpm::kTX} : mpm::TileRange{-1, 1 << 30};
    }
    float4 *cellf = nullptr;
    long long local_cells = 0;
    int plane_cap = 0;           // local planes the grids were allocated for (flowmpm_set_slab stays within)
    std::vector<float> shadow;   // host copy of records beyond `live` (single GPU)

    mpm::TiledState tiled{};     // bins, work items (FLOWMPM_ALGO_TILED)

    mpm::MigrationScratch migration{};   // device scratch of the leaver classification

    // Staging of the read-backs and uploads: one device buffer and one pinned host buffer per handle,
    // grown geometrically and freed in flowmpm_destroy, so that a renderer reading positions every
    // frame allocates nothing after its first frame.
    void *stage_dev = nullptr;
    size_t stage_dev_bytes = 0;
    void *stage_host = nullptr;
    size_t stage_host_bytes = 0;
    long long stage_allocations = 0;     // hipMalloc / hipHostMalloc calls made for the two buffers
    float *static_field = nullptr;       // this device's `scratch` array (flowmpm_download_field_static)

    bool profiling = false;      // the launches being enqueued right now are timed
    int profile_stride = 0;      // 0 off, k: every k-th step of flowmpm_step
    long long profile_step = 0;
    std::vector<EventPair> events;
    size_t events_used = 0;
    flowmpm_timings timings{};

    std::string err;
};

namespace {

int fail(flowmpm_sim *sim, int code, const std::string &msg)
{
    if (sim) sim->err = msg;
    g_last_error = msg;
    return code;
}

#define HIP_TRY(sim, expr)                                                            \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(sim, FLOWMPM_ERR_HIP,                                         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

#define CHECK_SIM(sim)                                                                \
    do {                                                                              \
        if (!(sim)) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");         \
        hipError_t d_ = hipSetDevice((sim)->cfg.device);                              \
        if (d_ != hipSuccess)                                                         \
            return fail(sim, FLOWMPM_ERR_HIP,                                         \
                        std::string("hipSetDevice: ") + hipGetErrorString(d_));       \
    } while (0)

inline unsigned blocks_for(long long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

void refresh_params(flowmpm_sim *s)
{
    mpm::Params &P = s->params;
    P.gx = s->cfg.grid_x; P.gy = s->cfg.grid_y; P.gz = s->cfg.grid_z;
    if (s->slab) {
        P.x_off = s->cfg.slab_x0 - 1;
        P.local_nx = s->cfg.slab_nx + 2;
    } else {
        P.x_off = 0;
        P.local_nx = s->cfg.grid_x;
    }
    const flowmpm_uniforms &u = s->uniforms;
    P.dt = u.dt; P.stiffness = u.stiffness; P.rest_density = u.rest_density;
    P.dynamic_viscosity = u.dynamic_viscosity; P.noise = u.noise; P.time = u.time;
    for (int a = 0; a < 3; a++) {
        P.ray_dir[a] = u.mouse_ray_direction[a];
        P.ray_origin[a] = u.mouse_ray_origin[a];
        P.mouse_force[a] = u.mouse_force[a];
    }
    P.cursor_interaction = u.cursor_interaction;
}

// ---- timing -----------------------------------------------------------------

// A pair of events for one timed launch, or two null events when profiling is off.  The
// events are attached to the kernel dispatch itself (hipExtLaunchKernelGGL), so they carry the
// kernel's own start and end timestamps and put no extra packet into the queue: timing leaves
// the schedule of the step as it is, and the durations are the ones a kernel trace reports.
struct TimedLaunch { hipEvent_t start = nullptr, stop = nullptr; };

int timing_events(flowmpm_sim *s, int slot, TimedLaunch *out, bool continuation = false)
{
    *out = TimedLaunch{};
    if (!s->profiling) return FLOWMPM_OK;
    if (s->events_used == s->events.size()) {
        EventPair p{};
        HIP_TRY(s, hipEventCreate(&p.start));
        HIP_TRY(s, hipEventCreate(&p.stop));
        s->events.push_back(p);
    }
    EventPair &p = s->events[s->events_used++];
    p.slot = slot;
    p.continuation = continuation;
    out->start = p.start;
    out->stop = p.stop;
    return FLOWMPM_OK;
}

int timing_collect(flowmpm_sim *s)
{
    if (s->events_used == 0) return FLOWMPM_OK;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < s->events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(s, hipEventElapsedTime(&ms, s->events[i].start, s->events[i].stop));
        s->timings.ms[s->events[i].slot] += ms;
        if (!s->events[i].continuation) s->timings.launches[s->events[i].slot] += 1;
    }
    s->events_used = 0;
    return FLOWMPM_OK;
}

// ---- the five stages ----------------------------------------------------------

int launch_check(flowmpm_sim *s, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(s, FLOWMPM_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return FLOWMPM_OK;
}

// float_grid = false: the fixed-point grid only.  The mass words that describe the float grid
// must survive until updateGrid has used them, so the step accumulates into the other buffer.
int stage_clear_grid(flowmpm_sim *s, bool float_grid = true)
{
    unsigned blocks = std::min<long long>(blocks_for(s->local_cells, 256), 256 * 16);
    if (!float_grid && s->mass_i == s->mass_of_cellf) {
        // the set that describes the float grid must survive until updateGrid: accumulate into the other one
        std::swap(s->mass_i, s->mass_spare);
        std::swap(s->cell, s->cell_spare);
        const bool clean = s->spare_clean;
        s->spare_clean = false;              // what is now the spare set is the dirty one
        if (clean) {                         // the last updateGrid has cleared it already: no launch
            s->tiles_marked = false;
            return FLOWMPM_OK;
        }
    }
    TimedLaunch t;
    int rc = timing_events(s, FLOWMPM_KERNEL_CLEAR_GRID, &t);
    if (rc) return rc;
    if (float_grid) {
        hipExtLaunchKernelGGL(mpm::clear_grid_kernel<true>, dim3(blocks), dim3(256), 0, s->stream, t.start, t.stop, 0,
                              s->cell, s->mass_i, s->cellf, s->mass_spare, s->local_cells);
        HIP_TRY(s, hipMemsetAsync(s->cell_spare, 0, (size_t)s->local_cells * sizeof(int4), s->stream));
        s->mass_of_cellf = s->mass_i;       // all zero, like the float grid
        s->spare_clean = true;
        if (s->dirty) HIP_TRY(s, hipMemsetAsync(s->dirty, 0, (size_t)s->tiled.n_tiles * sizeof(int), s->stream));
    } else {
        if (s->dirty)
            hipExtLaunchKernelGGL(mpm::clear_tiles_kernel, dim3(s->tiled.n_tiles), dim3(256), 0, s->stream, t.start,
                                  t.stop, 0, s->cell, s->mass_i, s->dirty, s->cell_bit(s->cell), s->mass_bit(s->mass_i),
                                  mpm::tile_dims(s->tiled), s->params, s->tile_range());
        else
            hipExtLaunchKernelGGL(mpm::clear_grid_kernel<false>, dim3(blocks), dim3(256), 0, s->stream, t.start,
                                  t.stop, 0, s->cell, s->mass_i, s->cellf, nullptr, s->local_cells);
    }
    s->tiles_marked = false;
    return launch_check(s, "clear_grid");
}

int stage_rebin_if_needed(flowmpm_sim *s, bool ride = false);
inline bool is_tiled(const flowmpm_sim *s);

// Before a p2g stage that sets no tile bits itself: if the stage that sets them has not run on
// these particles since the last clear, every tile counts as dirty.
int assume_all_tiles_dirty_unless_marked(flowmpm_sim *s)
{
    if (!s->dirty || s->tiles_marked) return FLOWMPM_OK;
    HIP_TRY(s, hipMemsetAsync(s->dirty, 0x0F, (size_t)s->tiled.n_tiles * sizeof(int), s->stream));
    s->tiles_marked = true;
    return FLOWMPM_OK;
}

int stage_p2g1(flowmpm_sim *s)
{
    int rc;
    if (is_tiled(s) && (rc = stage_rebin_if_needed(s))) return rc;
    if (s->live > 0) {
        TimedLaunch t;
        if ((rc = timing_events(s, FLOWMPM_KERNEL_P2G1, &t))) return rc;
        if (is_tiled(s)) {
            if ((rc = mpm::launch_p2g1_tiled(s->tiled, s->store, s->live, s->fixed(), s->params, s->stream, t.start,
                                             t.stop)))
                return fail(s, FLOWMPM_ERR_HIP, "p2g1_tiled launch failed");
        } else {
            hipExtLaunchKernelGGL(mpm::p2g1_scatter_kernel, dim3(blocks_for(s->live, 256)), dim3(256), 0, s->stream,
                                  t.start, t.stop, 0, s->store, s->live, s->fixed(), s->params);
        }
        if ((rc = launch_check(s, "p2g1"))) return rc;
        s->tiles_marked = true;
    }
    return FLOWMPM_OK;
}

int stage_p2g2(flowmpm_sim *s)
{
    int rc;
    if (is_tiled(s) && (rc = stage_rebin_if_needed(s))) return rc;
    if (s->live > 0) {
        if ((rc = assume_all_tiles_dirty_unless_marked(s))) return rc;
        TimedLaunch t;
        if ((rc = timing_events(s, FLOWMPM_KERNEL_P2G2, &t))) return rc;
        if (is_tiled(s)) {
            if ((rc = mpm::launch_p2g2_tiled(s->tiled, s->store, s->live, s->fixed(), s->params, s->stream, t.start,
                                             t.stop)))
                return fail(s, FLOWMPM_ERR_HIP, "p2g2_tiled launch failed");
        } else {
            hipExtLaunchKernelGGL(mpm::p2g2_scatter_kernel, dim3(blocks_for(s->live, 256)), dim3(256), 0, s->stream,
                                  t.start, t.stop, 0, s->store, s->live, s->fixed(), s->params);
        }
        if ((rc = launch_check(s, "p2g2"))) return rc;
    }
    return FLOWMPM_OK;
}

inline bool is_tiled(const flowmpm_sim *s) { return s->cfg.algorithm != FLOWMPM_ALGO_SCATTER; }

// The two-pass p2g of the tiled algorithm; `momentum` selects the pass.  stayers_only: the mass
// pass without the particles that have left the slab's owned planes.
// defer_scan: a re-bin that rides on a slab step counts its arrivals after this pass; the launch that builds the
// next item set then follows the arrivals' mass pass (stage_p2g_mass_arrivals).
int stage_p2g_pass(flowmpm_sim *s, bool momentum, bool stayers_only = false, int part = 0, bool defer_scan = false)
{
    int rc;
    if (!is_tiled(s)) return fail(s, FLOWMPM_ERR_INVALID, "the mass/momentum passes belong to the tiled algorithm");
    if (stayers_only && !s->slab) return fail(s, FLOWMPM_ERR_INVALID, "the stayers' mass pass is for slab handles");
    if (part != 0 && !s->slab) return fail(s, FLOWMPM_ERR_INVALID, "interior / boundary parts are for slab handles");
    if ((rc = stage_rebin_if_needed(s))) return rc;
    if (stayers_only) {
        s->stayers_done = true;
        s->arrivals_first = -1;
    }
    if (s->live > 0) {
        if (momentum && (rc = assume_all_tiles_dirty_unless_marked(s))) return rc;
        TimedLaunch t;
        if ((rc = timing_events(s, momentum ? FLOWMPM_KERNEL_P2G2 : FLOWMPM_KERNEL_P2G1, &t, part == 2))) return rc;
        const int keep_lo = stayers_only ? s->cfg.slab_x0 : -2147483647 - 1;
        const int keep_hi = stayers_only ? s->cfg.slab_x0 + s->cfg.slab_nx : 2147483647;
        rc = momentum ? mpm::launch_p2g_momentum_tiled(s->tiled, s->store, s->live, s->fixed(), s->params, s->stream,
                                                       t.start, t.stop, s->stage_part(part))
                      : mpm::launch_p2g_mass_tiled(s->tiled, s->store, s->live, s->fixed(), s->params, s->stream,
                                                   t.start, t.stop, keep_lo, keep_hi, defer_scan);
        if (rc) return fail(s, FLOWMPM_ERR_HIP, "p2g pass: could not reset the straggler counter");
        if ((rc = launch_check(s, momentum ? "p2g_momentum" : "p2g_mass"))) return rc;
        if (!momentum) s->tiles_marked = true;
    }
    return FLOWMPM_OK;
}

// The mass contributions of the particles inserted since the stayers' mass pass.
int stage_p2g_mass_arrivals(flowmpm_sim *s)
{
    if (!is_tiled(s) || !s->slab) return fail(s, FLOWMPM_ERR_INVALID, "the arrivals' mass pass is for tiled slab handles");
    if (!s->stayers_done) return fail(s, FLOWMPM_ERR_INVALID, "P2G_MASS_ARRIVALS without P2G_MASS_STAYERS");
    s->stayers_done = false;
    const long long first = s->arrivals_first;
    s->arrivals_first = -1;
    if (first >= 0 && first < s->live) {
        const long long count = s->live - first;
        hipLaunchKernelGGL(mpm::p2g_mass_range_kernel, dim3((unsigned)std::min<long long>(blocks_for(count, 256), 1024)),
                           dim3(256), 0, s->stream, s->store, (int)first, (int)s->live, s->fixed(), s->params,
                           mpm::tile_dims(s->tiled), s->tiled.ctl, s->tiled.key, s->tiled.rank, s->tiled.tile_count);
        int rc = launch_check(s, "p2g_mass_arrivals");
        if (rc) return rc;
    }
    mpm::launch_ride_scan(s->tiled, s->live, s->stream);       // of a riding re-bin whose scan was deferred to here
    s->tiles_marked = true;      // stayers and arrivals together are the resident particles
    return FLOWMPM_OK;
}

// clear_old: the caller is a whole step (flowmpm_step, the slab step), which no host call can interleave:
// the fixed-point set of the step before is zeroed on the way, and the next step starts without a clearGrid.
int stage_update_grid(flowmpm_sim *s, bool boundary_only = false, bool clear_old = false)
{
    if (boundary_only && !(s->slab && s->dirty))
        return fail(s, FLOWMPM_ERR_INVALID, "UPDATE_GRID_BOUNDARY is for tiled slab handles");
    TimedLaunch t;
    int rc = timing_events(s, FLOWMPM_KERNEL_UPDATE_GRID, &t, boundary_only);
    if (rc) return rc;
    unsigned blocks = std::min<long long>(blocks_for(s->local_cells, 256), 256 * 16);
    // (a workgroup per run of tiles along z)
    const int run = mpm::tile_run(s->tiled.n_tiles);
    const unsigned tile_blocks = (unsigned)(s->tiled.tiles_x * s->tiled.tiles_y * mpm::tile_runs_z(s->tiled.tiles_z, run));
    // the older set can be zeroed here when it is not the one being updated from
    const bool zero_old = clear_old && !boundary_only && s->dirty && s->mass_of_cellf != s->mass_i &&
                          s->mass_of_cellf == s->mass_spare;
    if (boundary_only) {
        // What the neighbours' momentum changed are the two pairs of planes the halo messages cover; every other
        // cell of the boundary tile columns would be recomputed from unchanged words.
        // (local planes 0, 1 and slab_nx, slab_nx + 1: halo_pair)
        const long long plane = (long long)s->cfg.grid_y * s->cfg.grid_z;
        const long long first_low = 0, first_high = (long long)s->cfg.slab_nx * plane, cells = 2 * plane;
        hipExtLaunchKernelGGL(mpm::update_grid_ranges_kernel, dim3(std::min<long long>(blocks_for(2 * cells, 256), 256 * 16)),
                              dim3(256), 0, s->stream, t.start, t.stop, 0, s->cell, s->mass_i, s->mass_of_cellf, s->cellf,
                              first_low, first_high, cells, s->params);
    } else if (s->dirty) {
        auto kernel = run == 1 ? mpm::update_grid_tiles_kernel<1> : mpm::update_grid_tiles_kernel<mpm::kTileRunMax>;
        hipExtLaunchKernelGGL(kernel, dim3(tile_blocks), dim3(256), 0, s->stream, t.start,
                              t.stop, 0, s->cell, s->mass_i, s->mass_of_cellf, s->cellf, s->dirty,
                              s->mass_bit(s->mass_i) | s->mass_bit(s->mass_of_cellf),
                              mpm::tile_dims(s->tiled), s->params, s->tile_range(), s->cell_spare,
                              zero_old ? (s->cell_bit(s->cell_spare) | s->mass_bit(s->mass_spare)) : 0);
    } else
        hipExtLaunchKernelGGL(mpm::update_grid_kernel, dim3(blocks), dim3(256), 0, s->stream, t.start, t.stop, 0,
                              s->cell, s->mass_i, s->mass_of_cellf, s->cellf, s->local_cells, s->params);
    if (s->mass_of_cellf != s->mass_i) {     // the float grid now follows this step's mass words
        s->mass_spare = s->mass_of_cellf;
        s->mass_of_cellf = s->mass_i;
    }
    if (zero_old) s->spare_clean = true;
    return launch_check(s, "update_grid");
}

// True when any uniform that selects a mode branch of g2p is set.
bool any_mode_on(const flowmpm_sim *s)
{
    const flowmpm_mode_uniforms &m = s->modes;
    bool on = s->uniforms.grid_mode != 0 || s->uniforms.front_gravity_mode != 0 ||
              s->uniforms.hide_percentage > 0.0f || m.cylinder_mode != 0 ||
              m.sphere_containment_mode != 0 || m.polygon_containment_mode != 0 || m.color_spheres_mode != 0 ||
              m.two_color_sphere_mode != 0;
    for (int k = 0; k < FLOWMPM_SEGMENTS; k++) on = on || m.segment_mode[k] != 0;
    return on;
}

// Stream-ordered refresh of the device copy: the record travels as a kernel argument, so no
// host buffer has to outlive the call.
__global__ void store_modes_kernel(mpm::ModeParams value, mpm::ModeParams *dst)
{
    const unsigned *src = reinterpret_cast<const unsigned *>(&value);
    unsigned *out = reinterpret_cast<unsigned *>(dst);
    for (unsigned i = threadIdx.x; i < sizeof(mpm::ModeParams) / sizeof(unsigned); i += blockDim.x) out[i] = src[i];
}

int stage_g2p(flowmpm_sim *s, int part = 0)
{
    int rc;
    if (part != 0 && !(s->slab && is_tiled(s)))
        return fail(s, FLOWMPM_ERR_INVALID, "interior / boundary parts are for tiled slab handles");
    if (is_tiled(s) && (rc = stage_rebin_if_needed(s))) return rc;
    s->migration.classified = -1;      // g2p moves the particles: a leaver classification made before it is void
    const mpm::ModeParams *modes = nullptr;
    if (any_mode_on(s)) {
        if (s->modes_stale || s->modes_for_live != s->live) {
            mpm::ModeParams value;
            value.u = s->modes;
            value.grid_mode = s->uniforms.grid_mode;
            value.front_gravity_mode = s->uniforms.front_gravity_mode;
            value.hide_percentage = s->uniforms.hide_percentage;
            // uniforms.numParticles (SIM:2513-2522): of the whole domain
            value.num_particles = (unsigned)(s->slab ? s->domain_particles : s->live);
            s->modes_for_live = s->live;
            hipLaunchKernelGGL(store_modes_kernel, dim3(1), dim3(256), 0, s->stream, value, s->d_modes);
            if ((rc = launch_check(s, "store_modes"))) return rc;
            s->modes_stale = false;
        }
        modes = s->d_modes;
    }
    if (s->live > 0) {
        TimedLaunch t;
        if ((rc = timing_events(s, FLOWMPM_KERNEL_G2P, &t, part == 2))) return rc;
        if (is_tiled(s)) {
            if ((rc = mpm::launch_g2p_tiled(s->tiled, s->store, s->live, s->cellf, s->params, modes, s->stream,
                                            t.start, t.stop, s->stage_part(part))))
                return fail(s, FLOWMPM_ERR_HIP, "g2p_tiled launch failed");
            if (part != 1) s->tiled.checked = false;      // the next step decides anew
            s->tiles_marked = false;       // the particles move
        } else if (modes) {
            hipExtLaunchKernelGGL(mpm::g2p_gather_kernel<true>, dim3(blocks_for(s->live, 256)), dim3(256), 0,
                                  s->stream, t.start, t.stop, 0, s->store, s->live, s->cellf, s->params, modes);
        } else {
            hipExtLaunchKernelGGL(mpm::g2p_gather_kernel<false>, dim3(blocks_for(s->live, 256)), dim3(256), 0,
                                  s->stream, t.start, t.stop, 0, s->store, s->live, s->cellf, s->params, modes);
        }
        if ((rc = launch_check(s, "g2p"))) return rc;
    }
    return FLOWMPM_OK;
}

// Tiled algorithm: particles must be grouped by grid tile before the first
// particle kernel of a step; regroup when the grouping is missing or stale.
// ride: the caller is flowmpm_step on the two-pass tiled algorithm, whose momentum pass and g2p stage
// follow in this step and complete a re-bin found due (mpm::rebin_if_due).
int stage_rebin_if_needed(flowmpm_sim *s, bool ride)
{
    mpm::TiledState &T = s->tiled;
    if (T.valid && T.checked) return FLOWMPM_OK;
    if (s->live == 0) return FLOWMPM_OK;
    TimedLaunch t;      // from the start of the first binning kernel to the end of the last
    int rc = timing_events(s, FLOWMPM_TIMING_REBIN, &t);
    if (rc) return rc;
    std::string msg;
    const int form = ride ? s->ride_rebins : 0;
    if (form == 0 || !T.valid) s->moved_since_find = true;       // the two-launch re-bin, if the device finds it due
    if (mpm::rebin_if_due(T, s->store, s->live, s->params, s->cfg.rebin_interval,
                          form ? s->drift_divisor : s->drift_divisor_standalone, s->stream, t.start, t.stop, &msg, form))
        return fail(s, FLOWMPM_ERR_HIP, "rebin: " + msg);
    return FLOWMPM_OK;
}

int validate_uniforms(flowmpm_sim *s, const flowmpm_uniforms *u)
{
    if (u->hide_percentage > 0.0f && s->slab && s->domain_particles <= 0)
        return fail(s, FLOWMPM_ERR_UNSUPPORTED,
                    "hidePercentage > 0 on a slab handle needs flowmpm_set_domain_particles first");
    if (u->cursor_interaction > 1u)
        return fail(s, FLOWMPM_ERR_INVALID, "cursorInteraction must be 0 or 1");
    return FLOWMPM_OK;
}

int alloc_store(flowmpm_sim *s, int which)
{
    s->store.capacity = s->capacity;
    size_t bytes = (size_t)mpm::kNumSlots * (size_t)s->capacity * sizeof(float);
    HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->store.base[which]), bytes));
    HIP_TRY(s, hipMemsetAsync(s->store.base[which], 0, bytes, s->stream));
    return FLOWMPM_OK;
}

// The handle's device staging buffer holds at most this much: larger transfers (whole-store parity dumps,
// the grids of a 512^3 domain) pass through it in chunks, so that one read-back never pins gigabytes of
// device memory until flowmpm_destroy.
constexpr size_t kDeviceStageMax = (size_t)256 << 20;

// The handle's device staging buffer, min(bytes, kDeviceStageMax) large or more; *got receives its size.
// Growing it frees the old one, which is safe here: every user synchronises the stream before it returns.
// Small buffers (the per-frame reads of a renderer) grow geometrically, large ones to what is asked for.
int stage_device(flowmpm_sim *s, size_t bytes, void **out, size_t *got = nullptr)
{
    const size_t need = std::min(bytes, kDeviceStageMax);
    if (s->stage_dev_bytes < need) {
        size_t want = need <= ((size_t)16 << 20) ? std::max(need, 2 * s->stage_dev_bytes) : need;
        want = std::min(kDeviceStageMax, (want + 255) / 256 * 256);
        HIP_TRY(s, hipStreamSynchronize(s->stream));
        if (s->stage_dev) (void)hipFree(s->stage_dev);
        s->stage_dev = nullptr;
        s->stage_dev_bytes = 0;
        HIP_TRY(s, hipMalloc(&s->stage_dev, want));
        s->stage_dev_bytes = want;
        s->stage_allocations += 1;
    }
    *out = s->stage_dev;
    if (got) *got = s->stage_dev_bytes;
    return FLOWMPM_OK;
}

// Transfers up to this size go through the handle's pinned host buffer (the per-frame reads of a
// renderer: 262,144 positions are 3 MiB); larger ones (whole-store parity dumps) go straight to the
// caller's memory.
constexpr size_t kPinnedStageMax = (size_t)64 << 20;

int stage_pinned(flowmpm_sim *s, size_t bytes, void **out)
{
    if (s->stage_host_bytes < bytes) {
        const size_t want = std::min(kPinnedStageMax, (std::max(bytes, 2 * s->stage_host_bytes) + 4095) / 4096 * 4096);
        HIP_TRY(s, hipStreamSynchronize(s->stream));
        if (s->stage_host) (void)hipHostFree(s->stage_host);
        s->stage_host = nullptr;
        s->stage_host_bytes = 0;
        HIP_TRY(s, hipHostMalloc(&s->stage_host, want, hipHostMallocDefault));
        s->stage_host_bytes = want;
        s->stage_allocations += 1;
    }
    *out = s->stage_host;
    return FLOWMPM_OK;
}

// Device -> caller's host memory, then waits for the stream.
int copy_out_and_wait(flowmpm_sim *s, void *host_dst, const void *dev_src, size_t bytes, const char *what)
{
    void *via = nullptr;
    if (bytes <= kPinnedStageMax) {
        int rc = stage_pinned(s, bytes, &via);
        if (rc) return rc;
    }
    hipError_t e = hipMemcpyAsync(via ? via : host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) return fail(s, FLOWMPM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    if (via) memcpy(host_dst, via, bytes);
    return FLOWMPM_OK;
}

// Uploads `count` host records into slots [first_slot, first_slot + count), one staging buffer's worth
// at a time.
int upload_records(flowmpm_sim *s, const float *aos, const int32_t *ids, long long first_slot,
                   long long count)
{
    if (count == 0) return FLOWMPM_OK;
    const size_t row_bytes = 28 * sizeof(float) + (ids ? sizeof(int) : 0);
    void *stage = nullptr;
    size_t got = 0;
    int rc = stage_device(s, (size_t)count * row_bytes, &stage, &got);
    if (rc) return rc;
    const long long rows_max = std::max<long long>(1, (long long)(got / row_bytes));
    for (long long lo = 0; lo < count; lo += rows_max) {
        const long long rows = std::min(rows_max, count - lo);
        const size_t aos_bytes = (size_t)rows * 28 * sizeof(float);
        float *d_aos = static_cast<float *>(stage);
        int *d_ids = ids ? reinterpret_cast<int *>(static_cast<char *>(stage) + aos_bytes) : nullptr;
        hipError_t e = hipMemcpyAsync(d_aos, aos + lo * 28, aos_bytes, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess && ids)
            e = hipMemcpyAsync(d_ids, ids + lo, (size_t)rows * sizeof(int), hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mpm::aos_to_soa_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, s->stream,
                               d_aos, d_ids, s->store, first_slot + lo, rows, ids ? 0LL : lo);
            e = hipGetLastError();
        }
        // the next chunk overwrites the staging buffer, and the caller's arrays are pageable
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) return fail(s, FLOWMPM_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
    }
    return FLOWMPM_OK;
}

// Positions must keep the 3x3x3 stencil inside the planes this handle holds.
int validate_positions(flowmpm_sim *s, const float *aos, long long count)
{
    const float lo_x = s->slab ? (float)s->cfg.slab_x0 : 1.0f;
    const float hi_x = s->slab ? (float)(s->cfg.slab_x0 + s->cfg.slab_nx) : (float)(s->cfg.grid_x - 1);
    const float hi_y = (float)(s->cfg.grid_y - 1), hi_z = (float)(s->cfg.grid_z - 1);
    for (long long i = 0; i < count; i++) {
        const float *p = aos + i * 28;
        bool ok = p[0] >= lo_x && p[0] < hi_x && p[1] >= 1.0f && p[1] < hi_y && p[2] >= 1.0f && p[2] < hi_z;
        if (!ok || !(p[0] >= 1.0f && p[0] < (float)(s->cfg.grid_x - 1))) {
            char buf[200];
            snprintf(buf, sizeof buf,
                     "particle %lld position (%g, %g, %g) puts its stencil outside the grid planes of this handle",
                     i, p[0], p[1], p[2]);
            return fail(s, FLOWMPM_ERR_INVALID, buf);
        }
    }
    return FLOWMPM_OK;
}

}  // namespace

extern "C" {

int flowmpm_version(void) { return 100; }

int flowmpm_device_count(int *count)
{
    if (!count) return fail(nullptr, FLOWMPM_ERR_INVALID, "null count");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, FLOWMPM_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    return FLOWMPM_OK;
}

const char *flowmpm_last_error(const flowmpm_sim *sim)
{
    return sim ? sim->err.c_str() : g_last_error.c_str();
}

int flowmpm_create(const flowmpm_config *config, flowmpm_sim **out)
{
    if (!config || !out) return fail(nullptr, FLOWMPM_ERR_INVALID, "null argument");
    *out = nullptr;
    const flowmpm_config &c = *config;
    if (c.grid_x < 6 || c.grid_y < 6 || c.grid_z < 6)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "grid must be at least 6 cells on every axis");
    if ((long long)c.grid_x * c.grid_y * c.grid_z > 2147483647LL / 4)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "grid too large for 32-bit cell indexing");
    // slot arrays are addressed with 32-bit byte offsets (mpm::Column)
    if (c.max_particles < 0 || c.max_particles > mpm::kMaxCapacity - 64)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "max_particles out of range (at most 2^30 - 64)");
    if (c.algorithm != FLOWMPM_ALGO_SCATTER && c.algorithm != FLOWMPM_ALGO_TILED &&
        c.algorithm != FLOWMPM_ALGO_TILED_UNFUSED)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "unknown algorithm");
    if (c.slab_nx < 0 || c.slab_x0 < 0 || c.slab_x0 + c.slab_nx > c.grid_x)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "slab outside the grid");
    if (c.slab_nx > 0 && c.slab_nx < 2)
        return fail(nullptr, FLOWMPM_ERR_INVALID, "a slab needs at least 2 planes");
    if (c.slab_max_nx < 0 || c.slab_max_nx > c.grid_x || (c.slab_nx == 0 && c.slab_max_nx != 0))
        return fail(nullptr, FLOWMPM_ERR_INVALID, "slab_max_nx must lie between 0 and grid_x (slab handles only)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, FLOWMPM_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (c.device < 0 || c.device >= ndev) return fail(nullptr, FLOWMPM_ERR_INVALID, "bad device ordinal");

    flowmpm_sim *s = new flowmpm_sim();
    s->cfg = c;
    if (s->cfg.rebin_interval < 1) s->cfg.rebin_interval = 1;
    s->slab = c.slab_nx > 0;
    s->capacity = std::max<long long>(c.max_particles, 1);
    // keep every slot array 256-byte aligned
    s->capacity = (s->capacity + 63) / 64 * 64;

    int rc = FLOWMPM_OK;
    auto body = [&]() -> int {
        HIP_TRY(s, hipSetDevice(c.device));
        HIP_TRY(s, scratch_setup(c.device, &s->static_field));
        if (c.external_stream) {
            s->stream = reinterpret_cast<hipStream_t>(c.stream);
        } else {
            HIP_TRY(s, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
            s->own_stream = true;
        }
        refresh_params(s);
        s->local_cells = (long long)s->params.local_nx * c.grid_y * c.grid_z;
        s->plane_cap = s->slab ? std::max(c.slab_nx, c.slab_max_nx) + 2 : c.grid_x;
        const size_t cells_cap = (size_t)s->plane_cap * c.grid_y * c.grid_z;
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->cell), cells_cap * sizeof(int4)));
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->cellf), cells_cap * sizeof(float4)));
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->mass_i), cells_cap * sizeof(int)));
        HIP_TRY(s, hipMemsetAsync(s->cell, 0, cells_cap * sizeof(int4), s->stream));
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->cell_spare), cells_cap * sizeof(int4)));
        HIP_TRY(s, hipMemsetAsync(s->cell_spare, 0, cells_cap * sizeof(int4), s->stream));
        s->cell_first = s->cell;
        s->spare_clean = true;
        HIP_TRY(s, hipMemsetAsync(s->mass_i, 0, cells_cap * sizeof(int), s->stream));
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->mass_spare), cells_cap * sizeof(int)));
        HIP_TRY(s, hipMemsetAsync(s->mass_spare, 0, cells_cap * sizeof(int), s->stream));
        s->mass_of_cellf = s->mass_i;
        s->mass_first = s->mass_i;
        HIP_TRY(s, hipMemsetAsync(s->cellf, 0, cells_cap * sizeof(float4), s->stream));
        HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->d_modes), sizeof(mpm::ModeParams)));
        HIP_TRY(s, hipMemsetAsync(s->d_modes, 0, sizeof(mpm::ModeParams), s->stream));
        int r = alloc_store(s, 0);
        if (r) return r;
        if (c.algorithm != FLOWMPM_ALGO_SCATTER) {
            if ((r = alloc_store(s, 1))) return r;
            std::string msg;
            if (mpm::tiled_init(s->tiled, s->params, s->capacity, s->plane_cap, s->stream, &msg))
                return fail(s, FLOWMPM_ERR_HIP, "tiled_init: " + msg);
            s->store.cur = &s->tiled.ctl->cur;
            // A slab takes in arrivals every step, all of them outside the binning until the next re-bin:
            // it pays to carry more of them for longer (measured on slabs of the 8-slab decomposition
            // of 2^27 particles: 256 -> 2.57, 96 -> 2.44, 48 -> 2.34, 24 -> 2.35, 12 -> 2.36 ms per step).
            if (s->slab) s->drift_divisor_standalone = 48;
            // flowmpm_step and the flowmpm_slab_step_* sequence let a re-bin ride; stage-by-stage hosts never do
            s->ride_rebins = c.algorithm == FLOWMPM_ALGO_TILED ? 2 : 0;
            s->migration.riding = &s->tiled.ctl->riding;
            s->migration.key = s->tiled.key;
            s->migration.rank = s->tiled.rank;
            // A riding re-bin costs a fraction of a standalone one, and the step it rides on still runs on the old
            // binning: it pays to start it earlier (steady state of the benchmark scene, noise uniform 0 / 1:
            // divisor 256 -> 10.17 / 7.78e9 per second, 512 -> 10.24 / 7.98, 1024 -> 10.22 / 7.98, 2048 -> 10.28 / 8.04).
            // With the count in the mass pass and the constant columns in the momentum pass (0.30 ms per re-bin):
            // 1024 -> 10.22 / 8.38, 4096 -> 10.31 / 8.21, 8192 -> 10.35 / 8.45, 16384 -> 10.37 / 8.48.
            // With the store order mixed by riding re-bins (the momentum pass and g2p of such a step read with a stride):
            // 2048 -> 12.01 / 9.68, 4096 -> 12.04 / 9.67, 8192 -> 11.99 / 9.51, 16384 -> 12.00 / 9.62.
            // A slab takes in a few thousand arrivals per step, all outside the binning until the next re-bin and
            // each costing three stages of global atomics: with re-bins this cheap it re-bins every few steps.
            if (s->ride_rebins) s->drift_divisor = s->slab ? kSlabRideDivisor : 4096;
            HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->dirty), (size_t)s->tiled.tiles_cap * sizeof(int)));
            HIP_TRY(s, hipMemsetAsync(s->dirty, 0, (size_t)s->tiled.tiles_cap * sizeof(int), s->stream));
        }
        // The leaver lists are sized once: a quarter of the slots may leave in one step before the
        // lists have to grow (growing frees device memory, which waits for the device).
        if (s->slab) s->migration.reserve = std::max<long long>(4096, s->capacity / 4);
        HIP_TRY(s, hipStreamSynchronize(s->stream));
        return FLOWMPM_OK;
    };
    rc = body();
    if (rc != FLOWMPM_OK) {
        g_last_error = s->err;
        flowmpm_destroy(s);
        return rc;
    }
    // reference defaults of the uniforms that have one (SIM:150)
    s->uniforms.dt = 0.1f;
    refresh_params(s);
    *out = s;
    return FLOWMPM_OK;
}

int flowmpm_destroy(flowmpm_sim *s)
{
    if (!s) return FLOWMPM_OK;
    (void)hipSetDevice(s->cfg.device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &p : s->events) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    mpm::tiled_free(s->tiled);
    if (s->store.base[0]) (void)hipFree(s->store.base[0]);
    if (s->store.base[1]) (void)hipFree(s->store.base[1]);
    if (s->cell) (void)hipFree(s->cell);
    if (s->cell_spare) (void)hipFree(s->cell_spare);
    if (s->mass_i) (void)hipFree(s->mass_i);
    if (s->mass_spare) (void)hipFree(s->mass_spare);
    if (s->dirty) (void)hipFree(s->dirty);
    if (s->cellf) (void)hipFree(s->cellf);
    if (s->d_modes) (void)hipFree(s->d_modes);
    if (s->migration.counters) (void)hipFree(s->migration.counters);
    if (s->migration.lists) (void)hipFree(s->migration.lists);
    if (s->stage_dev) (void)hipFree(s->stage_dev);
    if (s->stage_host) (void)hipHostFree(s->stage_host);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return FLOWMPM_OK;
}

int flowmpm_upload_particles(flowmpm_sim *s, const float *aos, const int32_t *ids, int64_t count)
{
    CHECK_SIM(s);
    if (count < 0 || (count > 0 && !aos)) return fail(s, FLOWMPM_ERR_INVALID, "bad particle array");
    if (count > s->cfg.max_particles)
        return fail(s, FLOWMPM_ERR_CAPACITY, "more particles than max_particles");
    if (!s->slab && ids) return fail(s, FLOWMPM_ERR_INVALID, "explicit ids are for slab handles only");
    int rc = validate_positions(s, aos, count);
    if (rc) return rc;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    if ((rc = upload_records(s, aos, ids, 0, count))) return rc;
    s->resident = s->live = s->uploaded = count;
    s->shadow.clear();
    s->tiled.valid = false;
    s->tiles_marked = false;
    s->tiled.listed = -1;         // the straggler list of the last momentum pass is stale
    return FLOWMPM_OK;
}

// Checks the arguments of a scene and fills the part of its description that does not depend on a handle.
static int scene_spec(flowmpm_sim *s, int32_t scene, uint32_t seed, int64_t first_id, int64_t count, double fill,
                      int gx, int gy, int gz, mpm::SceneSpec *out)
{
    if (scene != FLOWMPM_SCENE_ELLIPSOID && scene != FLOWMPM_SCENE_DAM_BREAK)
        return fail(s, FLOWMPM_ERR_INVALID, "unknown scene");
    if (count < 0 || first_id < 0 || first_id + count > 2147483647LL)
        return fail(s, FLOWMPM_ERR_INVALID, "particle ids must fit 31 bits");
    if (scene == FLOWMPM_SCENE_DAM_BREAK && !(fill > 0.0 && fill <= 1.0))
        return fail(s, FLOWMPM_ERR_INVALID, "the dam-break column needs 0 < fill <= 1");
    // SIM:90-112: the palette as hex strings, converted as the reference does (parseInt / 255)
    static const unsigned char palette[13][3] = {
        {0xFF, 0x56, 0xB7}, {0xBC, 0x96, 0xF9}, {0xC5, 0xF0, 0xF2}, {0x00, 0x74, 0x87}, {0x00, 0x74, 0x87},
        {0x15, 0x39, 0x43}, {0x23, 0x23, 0x56}, {0x50, 0x12, 0x1A}, {0xFF, 0x64, 0x25}, {0xD6, 0xF4, 0x99},
        {0x5C, 0xCA, 0xD8}, {0x5C, 0xCA, 0xD8}, {0xC5, 0xF0, 0xF2}};
    mpm::SceneSpec sp{};
    sp.kind = scene == FLOWMPM_SCENE_ELLIPSOID ? mpm::kSceneEllipsoid : mpm::kSceneDamBreak;
    sp.seed = seed;
    sp.first_id = first_id;
    sp.count = count;
    sp.gx = gx; sp.gy = gy; sp.gz = gz;
    sp.fill = fill;
    for (int c = 0; c < 13; c++)
        for (int k = 0; k < 3; k++) sp.palette[c][k] = (float)((double)palette[c][k] / 255.0);
    *out = sp;
    return FLOWMPM_OK;
}

// The host loop of init() (SIM:114-129) on the device; see kernels_scene.h.
int flowmpm_generate_particles(flowmpm_sim *s, int32_t scene, uint32_t seed, int64_t first_id, int64_t count,
                               double fill, int64_t *stored)
{
    CHECK_SIM(s);
    if (stored) *stored = 0;
    mpm::SceneSpec sp{};
    int rc0 = scene_spec(s, scene, seed, first_id, count, fill, s->cfg.grid_x, s->cfg.grid_y, s->cfg.grid_z, &sp);
    if (rc0) return rc0;
    if (!s->slab && first_id != 0)
        return fail(s, FLOWMPM_ERR_INVALID, "a single-domain handle holds the ids 0 .. count - 1");
    if (!s->slab && count > s->cfg.max_particles)
        return fail(s, FLOWMPM_ERR_CAPACITY, "more particles than max_particles");
    sp.keep_lo = s->cfg.slab_x0;
    sp.keep_hi = s->cfg.slab_x0 + s->cfg.slab_nx;
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    long long kept = count;
    if (count > 0) {
        int *counter = nullptr;
        if (s->slab) {
            void *stage = nullptr;
            int rc = stage_device(s, sizeof(int), &stage);
            if (rc) return rc;
            counter = static_cast<int *>(stage);
            HIP_TRY(s, hipMemsetAsync(counter, 0, sizeof(int), s->stream));
        }
        const unsigned blocks = (unsigned)std::min<long long>(blocks_for(count, 256), 16384);
        hipLaunchKernelGGL(mpm::generate_scene_kernel, dim3(blocks), dim3(256), 0, s->stream, sp, s->store, 0LL,
                           (long long)s->cfg.max_particles, counter);
        int rc = launch_check(s, "generate_scene");
        if (rc) return rc;
        if (counter) {
            int host_count = 0;
            if ((rc = copy_out_and_wait(s, &host_count, counter, sizeof(int), "generate_scene"))) return rc;
            kept = host_count;
        } else {
            HIP_TRY(s, hipStreamSynchronize(s->stream));
        }
    }
    s->shadow.clear();
    s->tiled.valid = false;
    s->tiles_marked = false;
    s->tiled.listed = -1;
    s->stayers_done = false;
    s->arrivals_first = -1;
    s->migration.classified = -1;
    if (kept > s->cfg.max_particles) {
        s->resident = s->live = s->uploaded = 0;
        char buf[160];
        snprintf(buf, sizeof buf, "%lld of the generated particles fall into this slab, max_particles is %lld",
                 kept, (long long)s->cfg.max_particles);
        return fail(s, FLOWMPM_ERR_CAPACITY, buf);
    }
    s->resident = s->live = s->uploaded = kept;
    if (stored) *stored = kept;
    return FLOWMPM_OK;
}

int flowmpm_plane_histogram(flowmpm_sim *s, int64_t *out, int32_t planes)
{
    CHECK_SIM(s);
    if (!out || planes != s->cfg.grid_x) return fail(s, FLOWMPM_ERR_INVALID, "the histogram has grid_x entries");
    if (planes > mpm::kHistPlanes) return fail(s, FLOWMPM_ERR_UNSUPPORTED, "plane histogram: grid_x above 8192");
    void *stage = nullptr;
    int rc = stage_device(s, (size_t)planes * sizeof(unsigned long long), &stage);
    if (rc) return rc;
    HIP_TRY(s, hipMemsetAsync(stage, 0, (size_t)planes * sizeof(unsigned long long), s->stream));
    if (s->resident > 0) {
        hipLaunchKernelGGL(mpm::plane_histogram_kernel, dim3((unsigned)std::min<long long>(blocks_for(s->resident, 256), 1024)),
                           dim3(256), 0, s->stream, s->store, (long long)s->resident, (int)planes,
                           static_cast<unsigned long long *>(stage));
        if ((rc = launch_check(s, "plane_histogram"))) return rc;
    }
    return copy_out_and_wait(s, out, stage, (size_t)planes * sizeof(int64_t), "plane_histogram");
}

int flowmpm_scene_plane_histogram(int32_t device, int32_t scene, uint32_t seed, int64_t first_id, int64_t count,
                                  const int32_t grid[3], double fill, int64_t *out)
{
    if (!grid || !out) return fail(nullptr, FLOWMPM_ERR_INVALID, "null argument");
    if (grid[0] < 6 || grid[1] < 6 || grid[2] < 6) return fail(nullptr, FLOWMPM_ERR_INVALID, "grid must be at least 6 cells on every axis");
    if (grid[0] > mpm::kHistPlanes) return fail(nullptr, FLOWMPM_ERR_UNSUPPORTED, "plane histogram: grid_x above 8192");
    mpm::SceneSpec sp{};
    int rc = scene_spec(nullptr, scene, seed, first_id, count, fill, grid[0], grid[1], grid[2], &sp);
    if (rc) return rc;
    HIP_TRY(nullptr, hipSetDevice(device));
    const size_t bytes = (size_t)grid[0] * sizeof(unsigned long long);
    unsigned long long *d = nullptr;
    HIP_TRY(nullptr, hipMalloc(reinterpret_cast<void **>(&d), bytes));
    hipError_t e = hipMemset(d, 0, bytes);
    if (e == hipSuccess && count > 0) {
        hipLaunchKernelGGL(mpm::scene_histogram_kernel, dim3((unsigned)std::min<long long>(blocks_for(count, 256), 2048)),
                           dim3(256), 0, nullptr, sp, (int)grid[0], d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, FLOWMPM_ERR_HIP, std::string("scene_plane_histogram: ") + hipGetErrorString(e));
    return FLOWMPM_OK;
}

int flowmpm_stencil_mass_sum(flowmpm_sim *s, int64_t *out)
{
    CHECK_SIM(s);
    if (!out) return fail(s, FLOWMPM_ERR_INVALID, "null output");
    void *stage = nullptr;
    int rc = stage_device(s, sizeof(unsigned long long), &stage);
    if (rc) return rc;
    HIP_TRY(s, hipMemsetAsync(stage, 0, sizeof(unsigned long long), s->stream));
    if (s->resident > 0) {
        hipLaunchKernelGGL(mpm::stencil_mass_sum_kernel, dim3(2048), dim3(256), 0, s->stream, s->store,
                           (long long)s->resident, static_cast<unsigned long long *>(stage));
        if ((rc = launch_check(s, "stencil_mass_sum"))) return rc;
    }
    return copy_out_and_wait(s, out, stage, sizeof(int64_t), "stencil_mass_sum");
}

int flowmpm_abi_sizes(int64_t sizes[4])
{
    if (!sizes) return fail(nullptr, FLOWMPM_ERR_INVALID, "null output");
    sizes[0] = (int64_t)sizeof(flowmpm_config);
    sizes[1] = (int64_t)sizeof(flowmpm_uniforms);
    sizes[2] = (int64_t)sizeof(flowmpm_mode_uniforms);
    sizes[3] = (int64_t)sizeof(flowmpm_timings);
    return FLOWMPM_OK;
}

int flowmpm_set_domain_particles(flowmpm_sim *s, int64_t count)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "set_domain_particles is for slab handles");
    if (count < 1 || count > 4294967295LL) return fail(s, FLOWMPM_ERR_INVALID, "domain particle count out of range");
    s->domain_particles = count;
    s->modes_stale = true;
    return FLOWMPM_OK;
}

int flowmpm_get_num_particles(flowmpm_sim *s, int64_t *live, int64_t *resident)
{
    if (!s) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");
    if (live) *live = s->live;
    if (resident) *resident = s->slab ? s->resident : s->uploaded;
    return FLOWMPM_OK;
}

int flowmpm_set_uniforms(flowmpm_sim *s, const flowmpm_uniforms *u)
{
    if (!s || !u) return fail(s, FLOWMPM_ERR_INVALID, "null argument");
    int rc = validate_uniforms(s, u);
    if (rc) return rc;
    if (u->grid_mode != s->uniforms.grid_mode || u->front_gravity_mode != s->uniforms.front_gravity_mode ||
        u->hide_percentage != s->uniforms.hide_percentage)
        s->modes_stale = true;
    s->uniforms = *u;
    refresh_params(s);
    return FLOWMPM_OK;
}

int flowmpm_set_option(flowmpm_sim *s, int32_t option, int64_t value)
{
    if (!s) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");
    if (!is_tiled(s)) return fail(s, FLOWMPM_ERR_INVALID, "the options belong to the tiled algorithms");
    switch (option) {
    case FLOWMPM_OPTION_RIDE_REBINS:
        if (value < 0 || value > 2) return fail(s, FLOWMPM_ERR_INVALID, "ride form must be 0, 1 or 2");
        if (value != 0 && s->cfg.algorithm != FLOWMPM_ALGO_TILED)
            return fail(s, FLOWMPM_ERR_UNSUPPORTED, "re-bins ride only on the two-pass tiled algorithm");
        if (value == 1 && s->slab)
            return fail(s, FLOWMPM_ERR_UNSUPPORTED, "a slab's riding re-bin counts in its mass pass (form 2)");
        s->ride_rebins = (int)value;
        return FLOWMPM_OK;
    case FLOWMPM_OPTION_DRIFT_DIVISOR:
        if (value < 0 || value > (1 << 30)) return fail(s, FLOWMPM_ERR_INVALID, "drift divisor out of range");
        s->drift_divisor = s->drift_divisor_standalone = (int)value;
        return FLOWMPM_OK;
    case FLOWMPM_OPTION_CHUNK:
        if (value < 1 || value > (1 << 30)) return fail(s, FLOWMPM_ERR_INVALID, "work-item size out of range");
        s->tiled.chunk_wanted = (int)value;
        s->tiled.valid = false;          // takes effect with a forced re-bin, which the next step performs
        return FLOWMPM_OK;
    case FLOWMPM_OPTION_SPARSE_TILE:
        if (value < 0 || value > (1 << 30)) return fail(s, FLOWMPM_ERR_INVALID, "sparse-tile bound out of range");
        s->tiled.sparse_wanted = (int)value;
        s->tiled.valid = false;
        return FLOWMPM_OK;
    default:
        return fail(s, FLOWMPM_ERR_INVALID, "unknown option");
    }
}

int flowmpm_set_mode_uniforms(flowmpm_sim *s, const flowmpm_mode_uniforms *m)
{
    if (!s || !m) return fail(s, FLOWMPM_ERR_INVALID, "null argument");
    if (m->polygon_vertex_count > 16u)
        return fail(s, FLOWMPM_ERR_INVALID, "polygonVertexCount above the 16 vertex uniforms");
    s->modes = *m;
    s->modes_stale = true;
    return FLOWMPM_OK;
}

int flowmpm_run_kernel(flowmpm_sim *s, int32_t kernel_id)
{
    CHECK_SIM(s);
    // clearGrid opens a step: the profiling stride counts steps here as it does in flowmpm_step
    if (kernel_id == FLOWMPM_KERNEL_CLEAR_GRID || kernel_id == FLOWMPM_KERNEL_CLEAR_FIXED_GRID)
        s->profiling = s->profile_stride > 0 && (s->profile_step++ % s->profile_stride) == 0;
    switch (kernel_id) {
    case FLOWMPM_KERNEL_CLEAR_GRID: return stage_clear_grid(s);
    case FLOWMPM_KERNEL_CLEAR_FIXED_GRID: return stage_clear_grid(s, false);
    case FLOWMPM_KERNEL_P2G1: return stage_p2g1(s);
    case FLOWMPM_KERNEL_P2G2: return stage_p2g2(s);
    case FLOWMPM_KERNEL_UPDATE_GRID: return stage_update_grid(s);
    case FLOWMPM_KERNEL_G2P: return stage_g2p(s);
    case FLOWMPM_KERNEL_P2G_MASS: return stage_p2g_pass(s, false);
    case FLOWMPM_KERNEL_P2G_MOMENTUM: return stage_p2g_pass(s, true);
    case FLOWMPM_KERNEL_P2G_MASS_STAYERS: return stage_p2g_pass(s, false, true);
    case FLOWMPM_KERNEL_P2G_MASS_ARRIVALS: return stage_p2g_mass_arrivals(s);
    case FLOWMPM_KERNEL_P2G_MOMENTUM_INTERIOR: return stage_p2g_pass(s, true, false, 1);
    case FLOWMPM_KERNEL_P2G_MOMENTUM_BOUNDARY: return stage_p2g_pass(s, true, false, 2);
    case FLOWMPM_KERNEL_G2P_INTERIOR: return stage_g2p(s, 1);
    case FLOWMPM_KERNEL_G2P_BOUNDARY: return stage_g2p(s, 2);
    case FLOWMPM_KERNEL_UPDATE_GRID_BOUNDARY: return stage_update_grid(s, true);
    default: return fail(s, FLOWMPM_ERR_INVALID, "unknown kernel id");
    }
}

int flowmpm_step(flowmpm_sim *s, int32_t n_steps, double dtime)
{
    CHECK_SIM(s);
    if (n_steps < 0) return fail(s, FLOWMPM_ERR_INVALID, "negative step count");
    const float time0 = s->uniforms.time;
    int rc = FLOWMPM_OK;
    for (int k = 0; k < n_steps && rc == FLOWMPM_OK; k++) {
        s->params.time = (float)((double)time0 + (double)k * dtime);
        s->profiling = s->profile_stride > 0 && (s->profile_step++ % s->profile_stride) == 0;
        if ((rc = stage_clear_grid(s, false))) break;
        if (s->cfg.algorithm == FLOWMPM_ALGO_TILED) {
            // the binning decision of the step; a re-bin rides on the passes below
            if (s->live > 0 && (rc = stage_rebin_if_needed(s, true))) break;
            if ((rc = stage_p2g_pass(s, false))) break;
            if ((rc = stage_p2g_pass(s, true))) break;
        } else {
            if ((rc = stage_p2g1(s))) break;
            if ((rc = stage_p2g2(s))) break;
        }
        if ((rc = stage_update_grid(s, false, true))) break;
        if ((rc = stage_g2p(s))) break;
        // bound the number of live event pairs
        if (s->profiling && s->events_used > 4096 && (rc = timing_collect(s))) break;
    }
    s->params.time = time0;
    s->profiling = s->profile_stride > 0;
    return rc;
}

int flowmpm_synchronize(flowmpm_sim *s)
{
    CHECK_SIM(s);
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    return FLOWMPM_OK;
}

int flowmpm_download_particles(flowmpm_sim *s, float *aos, int32_t *ids, int64_t capacity, int64_t *count)
{
    CHECK_SIM(s);
    const long long total = s->slab ? s->resident : s->uploaded;
    if (count) *count = total;
    if (!aos || capacity < total) return fail(s, FLOWMPM_ERR_INVALID, "output array too small");
    if (total == 0) return FLOWMPM_OK;
    const long long n = s->resident;
    if (n > 0) {
        // a window of rows at a time: rows are particle ids on a single-domain handle, slots on a slab
        const size_t row_bytes = 28 * sizeof(float) + (s->slab ? sizeof(int) : 0);
        void *stage = nullptr;
        size_t got = 0;
        int rc = stage_device(s, (size_t)n * row_bytes, &stage, &got);
        if (rc) return rc;
        const long long rows_max = std::max<long long>(1, (long long)(got / row_bytes));
        for (long long lo = 0; lo < n; lo += rows_max) {
            const long long rows = std::min(rows_max, n - lo);
            const size_t aos_bytes = (size_t)rows * 28 * sizeof(float);
            float *d_aos = static_cast<float *>(stage);
            int *d_ids = s->slab ? reinterpret_cast<int *>(static_cast<char *>(stage) + aos_bytes) : nullptr;
            hipLaunchKernelGGL(mpm::soa_to_aos_kernel, dim3(blocks_for(s->slab ? rows : n, 256)), dim3(256), 0, s->stream,
                               s->store, s->slab ? lo + rows : n, d_aos, d_ids, s->slab ? 0 : 1, lo, rows);
            if ((rc = launch_check(s, "soa_to_aos"))) return rc;
            if (s->slab && ids) {
                hipError_t e = hipMemcpyAsync(ids + lo, d_ids, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, s->stream);
                if (e != hipSuccess) return fail(s, FLOWMPM_ERR_HIP, std::string("download ids: ") + hipGetErrorString(e));
            }
            if ((rc = copy_out_and_wait(s, aos + lo * 28, d_aos, aos_bytes, "download"))) return rc;
        }
    }
    if (!s->slab && s->uploaded > s->resident)   // records beyond the live count are host-held
        memcpy(aos + s->resident * 28, s->shadow.data(), (size_t)(s->uploaded - s->resident) * 28 * sizeof(float));
    return FLOWMPM_OK;
}

int flowmpm_set_num_particles(flowmpm_sim *s, int64_t count)
{
    CHECK_SIM(s);
    if (s->slab) return fail(s, FLOWMPM_ERR_INVALID, "set_num_particles is for single-GPU handles");
    if (count < 0 || count > s->uploaded) return fail(s, FLOWMPM_ERR_INVALID, "count exceeds uploaded particles");
    if (count == s->live) return FLOWMPM_OK;
    // Bring every record back in id order, then re-upload the first `count`.
    std::vector<float> all((size_t)s->uploaded * 28);
    int64_t got = 0;
    int rc = flowmpm_download_particles(s, all.data(), nullptr, s->uploaded, &got);
    if (rc) return rc;
    if ((rc = upload_records(s, all.data(), nullptr, 0, count))) return rc;
    s->resident = s->live = count;
    s->shadow.assign(all.begin() + (size_t)count * 28, all.end());
    s->tiled.valid = false;
    s->tiles_marked = false;
    s->tiled.listed = -1;         // the straggler list of the last momentum pass is stale
    return FLOWMPM_OK;
}

static int field_columns(flowmpm_sim *s, int32_t field, int *slot, int *width)
{
    switch (field) {
    case FLOWMPM_FIELD_POSITION: *slot = mpm::kPosX; *width = 3; break;
    case FLOWMPM_FIELD_VELOCITY: *slot = mpm::kVelX; *width = 3; break;
    case FLOWMPM_FIELD_DENSITY: *slot = mpm::kDensity; *width = 1; break;
    case FLOWMPM_FIELD_COLOR: *slot = mpm::kColR; *width = 3; break;
    case FLOWMPM_FIELD_DIRECTION: *slot = mpm::kDirX; *width = 3; break;
    case FLOWMPM_FIELD_MASS: *slot = mpm::kMass; *width = 1; break;
    default: return fail(s, FLOWMPM_ERR_INVALID, "unknown field");
    }
    return FLOWMPM_OK;
}

int flowmpm_download_field(flowmpm_sim *s, int32_t field, float *out, int64_t capacity)
{
    CHECK_SIM(s);
    if (s->slab) return fail(s, FLOWMPM_ERR_INVALID, "download_field is for single-GPU handles");
    int slot, width;
    int rc = field_columns(s, field, &slot, &width);
    if (rc) return rc;
    const long long n = s->resident;
    if (!out || capacity < n) return fail(s, FLOWMPM_ERR_INVALID, "output array too small");
    if (n == 0) return FLOWMPM_OK;
    void *stage = nullptr;
    size_t got = 0;
    if ((rc = stage_device(s, (size_t)n * width * sizeof(float), &stage, &got))) return rc;
    const long long ids_max = std::max<long long>(1, (long long)(got / (width * sizeof(float))));
    for (long long lo = 0; lo < n; lo += ids_max) {
        const long long ids = std::min(ids_max, n - lo);
        hipLaunchKernelGGL(mpm::gather_field_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s->stream, s->store, n,
                           slot, width, static_cast<float *>(stage), lo, ids);
        if ((rc = launch_check(s, "gather_field"))) return rc;
        if ((rc = copy_out_and_wait(s, out + lo * width, stage, (size_t)ids * width * sizeof(float), "download_field")))
            return rc;
    }
    return FLOWMPM_OK;
}

// flowmpm_download_field without any allocation: the id-ordered values are staged in this device's
// `scratch` array (s->static_field, never the array's name) instead of the handle's staging buffer.
int flowmpm_download_field_static(flowmpm_sim *s, int32_t field, float *out, int64_t capacity)
{
    CHECK_SIM(s);
    if (s->slab) return fail(s, FLOWMPM_ERR_INVALID, "download_field_static is for single-GPU handles");
    int slot, width;
    int rc = field_columns(s, field, &slot, &width);
    if (rc) return rc;
    const long long n = s->resident;
    if (n > kStaticFieldParticles) {
        char buf[160];
        snprintf(buf, sizeof buf, "download_field_static holds at most %lld particles, the handle has %lld live: "
                                  "use flowmpm_download_field", kStaticFieldParticles, n);
        return fail(s, FLOWMPM_ERR_CAPACITY, buf);
    }
    if (!out || capacity < n) return fail(s, FLOWMPM_ERR_INVALID, "output array too small");
    if (n == 0) return FLOWMPM_OK;
    float *stage = s->static_field;
    if (!stage) return fail(s, FLOWMPM_ERR_HIP, "the static staging array was not set up on this device");
    hipLaunchKernelGGL(mpm::gather_field_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s->stream, s->store, n,
                       slot, width, stage, 0LL, n);
    if ((rc = launch_check(s, "gather_field"))) return rc;
    HIP_TRY(s, hipMemcpyAsync(out, stage, (size_t)n * width * sizeof(float), hipMemcpyDefault, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    return FLOWMPM_OK;
}

int flowmpm_get_local_grid(flowmpm_sim *s, int32_t *x_first, int32_t *x_planes)
{
    if (!s) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");
    if (x_first) *x_first = s->params.x_off;
    if (x_planes) *x_planes = s->params.local_nx;
    return FLOWMPM_OK;
}

int flowmpm_download_grid_i32(flowmpm_sim *s, int32_t *out, int64_t capacity_cells)
{
    CHECK_SIM(s);
    if (!out || capacity_cells < s->local_cells) return fail(s, FLOWMPM_ERR_INVALID, "output array too small");
    // The device keeps the mass words apart; hosts get the reference's {x, y, z, mass} records.
    void *stage = nullptr;
    size_t got = 0;
    int rc = stage_device(s, (size_t)s->local_cells * sizeof(int4), &stage, &got);
    if (rc) return rc;
    const long long cells_max = std::max<long long>(1, (long long)(got / sizeof(int4)));
    for (long long lo = 0; lo < s->local_cells; lo += cells_max) {
        const long long cells = std::min(cells_max, s->local_cells - lo);
        hipLaunchKernelGGL(mpm::assemble_cells_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s->stream,
                           s->cell + lo, s->mass_i + lo, static_cast<int4 *>(stage), cells);
        if ((rc = launch_check(s, "assemble_cells"))) return rc;
        if ((rc = copy_out_and_wait(s, out + lo * 4, stage, (size_t)cells * sizeof(int4), "download_grid_i32"))) return rc;
    }
    return FLOWMPM_OK;
}

int flowmpm_download_grid_f32(flowmpm_sim *s, float *out, int64_t capacity_cells)
{
    CHECK_SIM(s);
    if (!out || capacity_cells < s->local_cells) return fail(s, FLOWMPM_ERR_INVALID, "output array too small");
    return copy_out_and_wait(s, out, s->cellf, (size_t)s->local_cells * sizeof(float4), "download_grid_f32");
}

int flowmpm_set_profiling(flowmpm_sim *s, int32_t every_nth_step)
{
    CHECK_SIM(s);
    if (every_nth_step < 0) return fail(s, FLOWMPM_ERR_INVALID, "negative sampling stride");
    int rc = timing_collect(s);
    if (rc) return rc;
    s->profile_stride = every_nth_step;
    s->profile_step = 0;
    s->profiling = every_nth_step != 0;
    return FLOWMPM_OK;
}

int flowmpm_get_timings(flowmpm_sim *s, flowmpm_timings *out)
{
    CHECK_SIM(s);
    if (!out) return fail(s, FLOWMPM_ERR_INVALID, "null output");
    int rc = timing_collect(s);
    if (rc) return rc;
    *out = s->timings;
    if (is_tiled(s)) {
        mpm::RebinCtl ctl;
        HIP_TRY(s, hipMemcpyAsync(&ctl, s->tiled.ctl, sizeof ctl, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(s, hipStreamSynchronize(s->stream));
        out->launches[FLOWMPM_TIMING_REBINS_DONE] = ctl.rebins;
        out->launches[FLOWMPM_TIMING_STRAGGLERS] = (int64_t)ctl.stragglers;
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->cfg.device) == hipSuccess && khz > 0)
            out->ms[FLOWMPM_TIMING_REBINS_DONE] = (double)ctl.busy_ticks / (double)khz;
    }
    return FLOWMPM_OK;
}

int flowmpm_reset_timings(flowmpm_sim *s)
{
    CHECK_SIM(s);
    int rc = timing_collect(s);
    if (rc) return rc;
    s->timings = flowmpm_timings{};
    if (is_tiled(s))
        HIP_TRY(s, hipMemsetAsync(&s->tiled.ctl->rebins, 0, offsetof(mpm::RebinCtl, t_start) - offsetof(mpm::RebinCtl, rebins),
                                  s->stream));
    return FLOWMPM_OK;
}

int flowmpm_get_allocation_count(flowmpm_sim *s, int64_t *count)
{
    if (!s || !count) return fail(s, FLOWMPM_ERR_INVALID, "null argument");
    *count = s->stage_allocations + s->migration.allocations;
    return FLOWMPM_OK;
}

#if FLOWMPM_STAMPS
// Diagnostic build only: copies the clock stamps of `kernel` (0 momentum pass, 1 g2p) to the host,
// `slots` pairs of {shader cycles, 100 MHz ticks}.
int flowmpm_debug_clock_stamps(flowmpm_sim *s, int32_t kernel, uint64_t *out, int64_t slots)
{
    CHECK_SIM(s);
    if (kernel < 0 || kernel > 1 || !out || slots < 1 || slots > mpm::kStampSlots)
        return fail(s, FLOWMPM_ERR_INVALID, "bad stamp request");
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    HIP_TRY(s, hipMemcpyFromSymbol(out, HIP_SYMBOL(mpm::g_clock_stamps), (size_t)slots * 2 * sizeof(uint64_t),
                                   (size_t)kernel * mpm::kStampSlots * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FLOWMPM_OK;
}

// `slots` records of {set-up, particle loop, flush (100 MHz ticks), particles} per workgroup of `kernel`.
int flowmpm_debug_phase_stamps(flowmpm_sim *s, int32_t kernel, uint32_t *out, int64_t slots)
{
    CHECK_SIM(s);
    if (kernel < 0 || kernel > 1 || !out || slots < 1 || slots > mpm::kStampSlots)
        return fail(s, FLOWMPM_ERR_INVALID, "bad stamp request");
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    HIP_TRY(s, hipMemcpyFromSymbol(out, HIP_SYMBOL(mpm::g_phase_stamps), (size_t)slots * 4 * sizeof(uint32_t),
                                   (size_t)kernel * mpm::kStampSlots * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return FLOWMPM_OK;
}
#endif

// ---- slab decomposition --------------------------------------------------------

static int plane_index(flowmpm_sim *s, int32_t plane, long long *first_cell)
{
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    int lx;
    switch (plane) {
    case FLOWMPM_PLANE_LOW_HALO: lx = 0; break;
    case FLOWMPM_PLANE_LOW_OWNED: lx = 1; break;
    case FLOWMPM_PLANE_HIGH_OWNED: lx = s->cfg.slab_nx; break;
    case FLOWMPM_PLANE_HIGH_HALO: lx = s->cfg.slab_nx + 1; break;
    default: return fail(s, FLOWMPM_ERR_INVALID, "unknown plane");
    }
    *first_cell = (long long)lx * s->cfg.grid_y * s->cfg.grid_z;
    return FLOWMPM_OK;
}

int flowmpm_plane_bytes(flowmpm_sim *s, int64_t *bytes)
{
    if (!s) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (bytes) *bytes = (int64_t)s->cfg.grid_y * s->cfg.grid_z * (int64_t)sizeof(int4);
    return FLOWMPM_OK;
}

int flowmpm_plane_add_i32(flowmpm_sim *s, int32_t plane, const void *dev_src, uint32_t comp_mask)
{
    CHECK_SIM(s);
    long long first;
    int rc = plane_index(s, plane, &first);
    if (rc) return rc;
    if (!dev_src) return fail(s, FLOWMPM_ERR_INVALID, "null source");
    long long cells = (long long)s->cfg.grid_y * s->cfg.grid_z;
    hipLaunchKernelGGL(mpm::plane_add_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s->stream,
                       s->cell + first, s->mass_i + first, reinterpret_cast<const int4 *>(dev_src), cells,
                       comp_mask & 0xFu);
    return launch_check(s, "plane_add");
}

int flowmpm_plane_copy_out(flowmpm_sim *s, int32_t plane, void *dev_dst)
{
    CHECK_SIM(s);
    long long first;
    int rc = plane_index(s, plane, &first);
    if (rc) return rc;
    if (!dev_dst) return fail(s, FLOWMPM_ERR_INVALID, "null destination");
    long long cells = (long long)s->cfg.grid_y * s->cfg.grid_z;
    hipLaunchKernelGGL(mpm::assemble_cells_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s->stream,
                       s->cell + first, s->mass_i + first, reinterpret_cast<int4 *>(dev_dst), cells);
    return launch_check(s, "plane_copy_out");
}

int flowmpm_grid_sums(flowmpm_sim *s, int64_t *sums)
{
    CHECK_SIM(s);
    if (!sums) return fail(s, FLOWMPM_ERR_INVALID, "null output");
    const long long plane = (long long)s->cfg.grid_y * s->cfg.grid_z;
    const long long first = s->slab ? plane : 0;                       // skip the low halo plane
    const long long cells = s->slab ? (long long)s->cfg.slab_nx * plane : s->local_cells;
    void *stage = nullptr;
    int rc = stage_device(s, 4 * sizeof(unsigned long long), &stage);
    if (rc) return rc;
    unsigned long long *d = static_cast<unsigned long long *>(stage);
    HIP_TRY(s, hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), s->stream));
    hipLaunchKernelGGL(mpm::grid_sums_kernel, dim3(1024), dim3(256), 0, s->stream, s->cell + first,
                       s->mass_i + first, cells, d);
    if ((rc = launch_check(s, "grid_sums"))) return rc;
    return copy_out_and_wait(s, sums, d, 4 * sizeof(int64_t), "grid_sums");
}

// First cell and cell count of the two adjacent boundary planes of one side.
static int halo_pair(flowmpm_sim *s, int32_t side, uint32_t comp_mask, long long *first, long long *cells, int *k)
{
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (side != FLOWMPM_SIDE_LOW && side != FLOWMPM_SIDE_HIGH) return fail(s, FLOWMPM_ERR_INVALID, "unknown side");
    if ((comp_mask & 0xFu) == 0 || (comp_mask & ~0xFu)) return fail(s, FLOWMPM_ERR_INVALID, "bad component mask");
    const long long plane = (long long)s->cfg.grid_y * s->cfg.grid_z;
    *first = (side == FLOWMPM_SIDE_LOW ? 0 : (long long)s->cfg.slab_nx) * plane;
    *cells = 2 * plane;
    *k = __builtin_popcount(comp_mask & 0xFu);
    return FLOWMPM_OK;
}

int flowmpm_halo_words(flowmpm_sim *s, uint32_t comp_mask, int64_t *words)
{
    if (!s) return fail(nullptr, FLOWMPM_ERR_INVALID, "null handle");
    long long first, cells;
    int k;
    int rc = halo_pair(s, FLOWMPM_SIDE_LOW, comp_mask, &first, &cells, &k);
    if (rc) return rc;
    if (words) *words = cells * k;
    return FLOWMPM_OK;
}

int flowmpm_halo_pack(flowmpm_sim *s, int32_t side, uint32_t comp_mask, void *dev_dst)
{
    CHECK_SIM(s);
    long long first, cells;
    int k;
    int rc = halo_pair(s, side, comp_mask, &first, &cells, &k);
    if (rc) return rc;
    if (!dev_dst) return fail(s, FLOWMPM_ERR_INVALID, "null destination");
    hipLaunchKernelGGL(mpm::halo_pack_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s->stream, s->cell + first,
                       s->mass_i + first, reinterpret_cast<int *>(dev_dst), cells, comp_mask & 0xFu, k);
    return launch_check(s, "halo_pack");
}

int flowmpm_halo_add(flowmpm_sim *s, int32_t side, uint32_t comp_mask, const void *dev_src)
{
    CHECK_SIM(s);
    long long first, cells;
    int k;
    int rc = halo_pair(s, side, comp_mask, &first, &cells, &k);
    if (rc) return rc;
    if (!dev_src) return fail(s, FLOWMPM_ERR_INVALID, "null source");
    hipLaunchKernelGGL(mpm::halo_add_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, s->stream, s->cell + first,
                       s->mass_i + first, reinterpret_cast<const int *>(dev_src), cells, comp_mask & 0xFu, k);
    return launch_check(s, "halo_add");
}

int flowmpm_set_slab(flowmpm_sim *s, int32_t slab_x0, int32_t slab_nx)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (slab_nx < 2 || slab_x0 < 0 || slab_x0 + slab_nx > s->cfg.grid_x) return fail(s, FLOWMPM_ERR_INVALID, "slab outside the grid");
    if (slab_nx + 2 > s->plane_cap) {
        char buf[160];
        snprintf(buf, sizeof buf, "%d planes asked for, the handle was created for at most %d (flowmpm_config.slab_max_nx)",
                 slab_nx, s->plane_cap - 2);
        return fail(s, FLOWMPM_ERR_CAPACITY, buf);
    }
    if (s->slab_phase != 0) return fail(s, FLOWMPM_ERR_INVALID, "a slab is re-cut between steps, not inside one");
    if (slab_x0 == s->cfg.slab_x0 && slab_nx == s->cfg.slab_nx) return FLOWMPM_OK;
    s->cfg.slab_x0 = slab_x0;
    s->cfg.slab_nx = slab_nx;
    refresh_params(s);
    s->local_cells = (long long)s->params.local_nx * s->cfg.grid_y * s->cfg.grid_z;
    // the planes the local grids hold have changed: what they held means nothing any more
    HIP_TRY(s, hipMemsetAsync(s->cell, 0, (size_t)s->local_cells * sizeof(int4), s->stream));
    HIP_TRY(s, hipMemsetAsync(s->cell_spare, 0, (size_t)s->local_cells * sizeof(int4), s->stream));
    s->spare_clean = true;
    HIP_TRY(s, hipMemsetAsync(s->cellf, 0, (size_t)s->local_cells * sizeof(float4), s->stream));
    HIP_TRY(s, hipMemsetAsync(s->mass_i, 0, (size_t)s->local_cells * sizeof(int), s->stream));
    HIP_TRY(s, hipMemsetAsync(s->mass_spare, 0, (size_t)s->local_cells * sizeof(int), s->stream));
    s->mass_of_cellf = s->mass_i;
    if (is_tiled(s)) {
        mpm::tiled_set_planes(s->tiled, s->params, s->params.local_nx);
        HIP_TRY(s, hipMemsetAsync(s->dirty, 0, (size_t)s->tiled.tiles_cap * sizeof(int), s->stream));
        s->tiled.valid = false;          // the next step bins the particles under the new tiles
        s->tiled.checked = false;
        s->tiled.listed = -1;
        s->tiled.scan_after_mass = false;
    }
    s->tiles_marked = false;
    s->stayers_done = false;
    s->arrivals_first = -1;
    s->migration.classified = -1;
    return FLOWMPM_OK;
}

// ---- one slab step in six calls (see include/flowmpm.h) -----------------------------------

static int slab_step_enter(flowmpm_sim *s, int phase, const char *name)
{
    if (!s->slab || s->cfg.algorithm != FLOWMPM_ALGO_TILED)
        return fail(s, FLOWMPM_ERR_INVALID, std::string(name) + " is for slab handles of FLOWMPM_ALGO_TILED");
    if (s->slab_phase != phase) {
        static const char *const order[] = {"flowmpm_slab_step_begin", "flowmpm_slab_step_pack_mass", "flowmpm_slab_step_interior",
                                            "flowmpm_slab_step_boundary", "flowmpm_slab_step_grid", "flowmpm_slab_step_end"};
        const std::string want = order[s->slab_phase];
        s->slab_phase = 0;
        return fail(s, FLOWMPM_ERR_INVALID, std::string(name) + " out of order: " + want + " comes next");
    }
    return FLOWMPM_OK;
}

// Packs (add = false) or merges (add = true) the halo pairs of both sides in one launch; a null buffer
// means no neighbour on that side.  A thin slab's two pairs overlap (slab_nx < 2 cannot be created, at
// slab_nx = 2 they are disjoint), so the two sides of one launch never touch the same cell.
static int halo_sides(flowmpm_sim *s, bool add, uint32_t mask, const void *low, const void *high)
{
    if (!low && !high) return FLOWMPM_OK;
    long long first, cells;
    int k;
    int rc = halo_pair(s, FLOWMPM_SIDE_LOW, mask, &first, &cells, &k);
    if (rc) return rc;
    mpm::HaloSides h{};
    h.first[0] = first;
    h.buf[0] = static_cast<int *>(const_cast<void *>(low));
    if ((rc = halo_pair(s, FLOWMPM_SIDE_HIGH, mask, &first, &cells, &k))) return rc;
    h.first[1] = first;
    h.buf[1] = static_cast<int *>(const_cast<void *>(high));
    const dim3 grid(blocks_for(cells, 256), 2);
    if (add)
        hipLaunchKernelGGL(mpm::halo_add_sides_kernel, grid, dim3(256), 0, s->stream, s->cell, s->mass_i, h, cells, mask & 0xFu, k);
    else
        hipLaunchKernelGGL(mpm::halo_pack_sides_kernel, grid, dim3(256), 0, s->stream, s->cell, s->mass_i, h, cells, mask & 0xFu, k);
    return launch_check(s, add ? "halo_add" : "halo_pack");
}

int flowmpm_slab_step_begin(flowmpm_sim *s)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 0, "flowmpm_slab_step_begin");
    if (rc) return rc;
    s->profiling = s->profile_stride > 0 && (s->profile_step++ % s->profile_stride) == 0;
    if ((rc = stage_clear_grid(s, false))) return rc;
    // the binning decision of the step; a re-bin rides on the passes that follow
    if (s->live > 0 && (rc = stage_rebin_if_needed(s, true))) return rc;
    if ((rc = stage_p2g_pass(s, false, true, 0, true))) return rc;
    s->slab_phase = 1;
    return FLOWMPM_OK;
}

int flowmpm_slab_step_pack_mass(flowmpm_sim *s, void *send_low, void *send_high)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 1, "flowmpm_slab_step_pack_mass");
    if (rc) return rc;
    s->slab_phase = 0;
    if ((rc = stage_p2g_mass_arrivals(s))) return rc;
    if ((rc = halo_sides(s, false, FLOWMPM_COMP_MASS, send_low, send_high))) return rc;
    s->slab_phase = 2;
    return FLOWMPM_OK;
}

int flowmpm_slab_step_interior(flowmpm_sim *s)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 2, "flowmpm_slab_step_interior");
    if (rc) return rc;
    s->slab_phase = 0;
    if ((rc = stage_p2g_pass(s, true, false, 1))) return rc;
    s->slab_phase = 3;
    return FLOWMPM_OK;
}

int flowmpm_slab_step_boundary(flowmpm_sim *s, const void *recv_low, const void *recv_high, void *send_low,
                               void *send_high)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 3, "flowmpm_slab_step_boundary");
    if (rc) return rc;
    s->slab_phase = 0;
    if ((rc = halo_sides(s, true, FLOWMPM_COMP_MASS, recv_low, recv_high))) return rc;
    if ((rc = stage_p2g_pass(s, true, false, 2))) return rc;
    if ((rc = halo_sides(s, false, FLOWMPM_COMP_MOMENTUM, send_low, send_high))) return rc;
    s->slab_phase = 4;
    return FLOWMPM_OK;
}

int flowmpm_slab_step_grid(flowmpm_sim *s)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 4, "flowmpm_slab_step_grid");
    if (rc) return rc;
    s->slab_phase = 0;
    if ((rc = stage_update_grid(s, false, true))) return rc;
    if ((rc = stage_g2p(s, 1))) return rc;
    s->slab_phase = 5;
    return FLOWMPM_OK;
}

int flowmpm_slab_step_end(flowmpm_sim *s, const void *recv_low, const void *recv_high, const int32_t *plane_starts,
                          int32_t n_ranks, void *dev_counts)
{
    CHECK_SIM(s);
    int rc = slab_step_enter(s, 5, "flowmpm_slab_step_end");
    if (rc) return rc;
    s->slab_phase = 0;
    if ((rc = halo_sides(s, true, FLOWMPM_COMP_MOMENTUM, recv_low, recv_high))) return rc;
    if (recv_low || recv_high) {
        if ((rc = stage_update_grid(s, true))) return rc;
    }
    if ((rc = stage_g2p(s, 2))) return rc;
    if (s->profiling && s->events_used > 4096 && (rc = timing_collect(s))) return rc;      // bound the live event pairs
    s->profiling = s->profile_stride > 0;
    if (plane_starts) return flowmpm_find_leavers(s, plane_starts, n_ranks, dev_counts);
    return FLOWMPM_OK;
}

int flowmpm_extract_leavers(flowmpm_sim *s, const int32_t *plane_starts, int32_t n_ranks, void *dev_out,
                            int64_t capacity_records, int64_t *counts)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (!plane_starts || n_ranks < 1 || n_ranks > mpm::kMaxRanks || !counts)
        return fail(s, FLOWMPM_ERR_INVALID, "bad rank table");
    std::string msg;
    long long new_resident = s->resident;
    int rc = mpm::extract_leavers(s->migration, s->store, s->resident, s->cfg.slab_x0, s->cfg.slab_nx, plane_starts,
                                  n_ranks, reinterpret_cast<float *>(dev_out), capacity_records, counts, s->stream,
                                  &new_resident, &msg);
    if (rc == 1) return fail(s, FLOWMPM_ERR_HIP, "extract_leavers: " + msg);
    if (rc == 2) return fail(s, FLOWMPM_ERR_CAPACITY, "extract_leavers: " + msg);
    if (rc) return fail(s, FLOWMPM_ERR_INVALID, "extract_leavers: " + msg);
    s->resident = s->live = new_resident;
    s->tiles_marked = false;
    s->tiled.listed = -1;         // the straggler list of the last momentum pass is stale
    return FLOWMPM_OK;
}

int flowmpm_find_leavers(flowmpm_sim *s, const int32_t *plane_starts, int32_t n_ranks, void *dev_counts)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (!plane_starts || n_ranks < 1 || n_ranks > mpm::kMaxRanks)
        return fail(s, FLOWMPM_ERR_INVALID, "bad rank table");
    std::string msg;
    // Without a count buffer, and with a classification of the same particles in flight, this is the
    // repeat before the pack: the device skips it unless a re-bin has moved the particles since.
    const bool repeat = !dev_counts && s->migration.classified == s->resident && s->migration.table.n == n_ranks;
    const int *epoch = s->tiled.ctl ? &s->tiled.ctl->epoch : nullptr;
    if (mpm::find_leavers(s->migration, s->store, s->resident, s->cfg.slab_x0, s->cfg.slab_nx, plane_starts, n_ranks,
                          reinterpret_cast<long long *>(dev_counts), epoch, repeat, s->stream, &msg))
        return fail(s, FLOWMPM_ERR_HIP, "find_leavers: " + msg);
    if (!repeat) s->moved_since_find = false;
    return FLOWMPM_OK;
}

int flowmpm_pack_leavers(flowmpm_sim *s, const int64_t *counts, int32_t n_ranks, void *dev_out,
                         int64_t capacity_records)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (!counts || n_ranks < 1 || n_ranks > mpm::kMaxRanks) return fail(s, FLOWMPM_ERR_INVALID, "bad counts");
    for (int r = 0; r < n_ranks; r++)
        if (counts[r] < 0 || counts[r] > s->resident) return fail(s, FLOWMPM_ERR_INVALID, "bad counts");
    std::string msg;
    // A re-bin between the classification and this call has moved the particles: the device repeats the
    // classification on the slots they occupy now, if and only if that happened (the store's binning epoch).
    if (s->tiled.ctl && s->moved_since_find && s->migration.classified == s->resident && s->migration.table.n == n_ranks) {
        const mpm::RankTable table = s->migration.table;        // find_leavers rewrites the scratch's copy
        if (mpm::find_leavers(s->migration, s->store, s->resident, s->cfg.slab_x0, s->cfg.slab_nx, table.starts, n_ranks,
                              nullptr, &s->tiled.ctl->epoch, true, s->stream, &msg))
            return fail(s, FLOWMPM_ERR_HIP, "pack_leavers: " + msg);
    }
    long long new_resident = s->resident;
    int rc = mpm::pack_leavers(s->migration, s->store, s->resident, s->cfg.slab_x0, s->cfg.slab_nx, counts, n_ranks,
                               reinterpret_cast<float *>(dev_out), capacity_records, s->stream, &new_resident, &msg);
    if (rc == 1) return fail(s, FLOWMPM_ERR_HIP, "pack_leavers: " + msg);
    if (rc == 2) return fail(s, FLOWMPM_ERR_CAPACITY, "pack_leavers: " + msg);
    if (rc) return fail(s, FLOWMPM_ERR_INVALID, "pack_leavers: " + msg);
    s->resident = s->live = new_resident;
    s->tiles_marked = false;
    s->tiled.listed = -1;         // the straggler list of the last momentum pass is stale
    return FLOWMPM_OK;
}

int flowmpm_insert_arrivals(flowmpm_sim *s, const void *dev_in, int64_t count)
{
    CHECK_SIM(s);
    if (!s->slab) return fail(s, FLOWMPM_ERR_INVALID, "not a slab handle");
    if (count < 0 || (count > 0 && !dev_in)) return fail(s, FLOWMPM_ERR_INVALID, "bad arrival buffer");
    if (count == 0) return FLOWMPM_OK;
    if (s->resident + count > s->cfg.max_particles)
        return fail(s, FLOWMPM_ERR_CAPACITY, "arrivals exceed max_particles of this slab");
    hipLaunchKernelGGL(mpm::unpack_records_kernel, dim3(blocks_for(count, 256)), dim3(256), 0, s->stream,
                       reinterpret_cast<const float *>(dev_in), s->store, (long long)s->resident, (long long)count);
    int rc = launch_check(s, "insert_arrivals");
    if (rc) return rc;
    if (s->stayers_done && s->arrivals_first < 0) s->arrivals_first = s->resident;
    s->resident += count;
    s->live = s->resident;      // arrivals sit past the binned range until the next re-bin
    s->tiles_marked = false;
    s->tiled.listed = -1;         // the straggler list of the last momentum pass is stale
    return FLOWMPM_OK;
}

}  // extern "C"

#include "slab_group.h"
