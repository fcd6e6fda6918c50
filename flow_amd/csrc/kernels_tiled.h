// Tiled particle<->grid transfers for CDNA4 (FLOWMPM_ALGO_TILED).
//
// Particles are kept grouped ("binned") by the grid tile that contains
// floor(position).  One workgroup owns one work item = a run of particles of one
// tile, and keeps that tile plus a one-cell rim (the "arena") in LDS:
//
//   p2g1 / p2g2: every stencil contribution is encoded to the reference's
//     fixed-point i32 FIRST (SIM:279-281) and then summed with LDS integer
//     atomics; the arena is flushed to the global grid with one i32 atomic per
//     touched cell component.  Integer addition is associative and wraps, so
//     the grid is bit-identical to the reference's 27x4 global atomics per
//     particle, whatever the grouping.
//   g2p: the float grid of the arena is staged in LDS once and gathered from there.
//
// Binning may be stale by a few steps (rebin_interval): a contribution whose
// cell fell outside the arena goes straight to the global grid, so staleness
// costs speed, never correctness.
#pragma once

#include <cstddef>
#include <string>

#include "kernels_migrate.h"

namespace mpm {

// Tile shape in cells.
#ifndef FLOWMPM_TX
#define FLOWMPM_TX 8
#define FLOWMPM_TY 8
#define FLOWMPM_TZ 8
#endif
constexpr int kTX = FLOWMPM_TX, kTY = FLOWMPM_TY, kTZ = FLOWMPM_TZ;
// Rim of the arena in cells: 1 is what the stencil of a freshly binned particle needs; every
// cell beyond that is drift a stale bin tolerates before its particle falls back to global atomics.
#ifndef FLOWMPM_RIM
#define FLOWMPM_RIM 2
#endif
constexpr int kRim = FLOWMPM_RIM;
constexpr int kAX = kTX + 2 * kRim, kAY = kTY + 2 * kRim, kAZ = kTZ + 2 * kRim;
constexpr int kArenaCells = kAX * kAY * kAZ;
constexpr int kArenaPad = kArenaCells + 8;     // component stride in LDS (dwords)
// Most particles one work item (= one workgroup's run of particles of one tile) holds.  Chosen per
// binning from the particle count (choose_chunk): a large scene gets the upper bound, where the arena's
// set-up and flush are amortised best; a small one gets more, smaller items, so that the chip is filled
// (several workgroups per tile merge through the same integer atomics: bits do not change).
#ifndef FLOWMPM_TILED_THREADS
#define FLOWMPM_TILED_THREADS 256
#endif
constexpr int kTiledThreads = FLOWMPM_TILED_THREADS;
constexpr int kChunkMax = 16 * kTiledThreads, kChunkMin = kTiledThreads;
constexpr int kLargeItem = 1024;     // particles; see scan_tiles
#ifndef FLOWMPM_MIX_EVERY
#define FLOWMPM_MIX_EVERY 4
#endif
constexpr int kMixEvery = FLOWMPM_MIX_EVERY;      // riding re-bins per mixing one (p2g_mass_tiled_kernel)
// Minimum waves per SIMD the register allocator must leave room for (second argument of
// __launch_bounds__) in the two register-heavy kernels; measured, not guessed.
#ifndef FLOWMPM_P2G_WAVES
#define FLOWMPM_P2G_WAVES 1
#endif
#ifndef FLOWMPM_G2P_WAVES
#define FLOWMPM_G2P_WAVES 1
#endif

// Diagnostic build only (-DFLOWMPM_STAMPS=1, tools/clock_stamps.py): every workgroup of the momentum
// pass and of g2p stamps the shader clock (s_memtime, one tick per shader cycle) and the constant
// 100 MHz clock (s_memrealtime) around its particle loop; their quotient is the clock the chip held
// under this load.  The stamps go to an array nothing else reads; no stamp executes in a normal build.
#ifndef FLOWMPM_STAMPS
#define FLOWMPM_STAMPS 0
#endif
#if FLOWMPM_STAMPS
constexpr int kStampSlots = 1 << 15;
__device__ unsigned long long g_clock_stamps[2][kStampSlots][2];      // [kernel][workgroup][shader, real]
// phases of a workgroup in 100 MHz ticks: arena set-up, particle loop, flush; and its particle count
__device__ unsigned int g_phase_stamps[2][kStampSlots][4];
struct ClockStamp {
    unsigned long long t0, r0, enter;
    __device__ __forceinline__ void entered()
    {
        if (threadIdx.x == 0) enter = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void begin()
    {
        if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ __forceinline__ void end(int kernel)
    {
        if (threadIdx.x == 0 && blockIdx.x < (unsigned)kStampSlots) {
            g_clock_stamps[kernel][blockIdx.x][0] = __builtin_amdgcn_s_memtime() - t0;
            g_clock_stamps[kernel][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
    }
    __device__ __forceinline__ void left(int kernel, int particles)
    {
        if (threadIdx.x == 0 && blockIdx.x < (unsigned)kStampSlots) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            g_phase_stamps[kernel][blockIdx.x][0] = (unsigned)(r0 - enter);
            g_phase_stamps[kernel][blockIdx.x][1] = (unsigned)g_clock_stamps[kernel][blockIdx.x][1];
            g_phase_stamps[kernel][blockIdx.x][2] = (unsigned)(now - r0) - (unsigned)g_clock_stamps[kernel][blockIdx.x][1];
            g_phase_stamps[kernel][blockIdx.x][3] = (unsigned)particles;
        }
    }
};
#else
struct ClockStamp {
    __device__ __forceinline__ void entered() {}
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void end(int) {}
    __device__ __forceinline__ void left(int, int) {}
};
#endif

// Particles whose stencil no longer fits their bin's arena (they drifted since the last
// re-bin) are not handled by the tiled kernels: their slot index is appended here and a
// small follow-up kernel sends their contributions straight to the global grid.  Keeping
// that rare path out of the hot kernels keeps their register allocation small.
struct Stragglers {
    int *count;       // device scalar: zero when a stage starts
    int *next_count;  // the counter of the next stage, zeroed by this stage's follow-up kernel
    int *list;        // slot indices, capacity = store capacity
};

// Re-binning control words, on the device.  The decision to re-bin is taken there, from the
// number of particles that sit outside their bin's arena, so that the host can enqueue any
// number of steps ahead and a burst of fast particles still triggers a re-bin at once.
struct alignas(32) RebinCtl {
    // The head: what every workgroup of a tiled kernel needs before it can start, laid out so that it
    // arrives with ONE 32-byte scalar load (CtlHead below; read one after the other -- the current store,
    // then the item set, then its item count -- these words were three dependent round trips).
    int cur;         // which of the two stores is current (StoreSel::cur points here)
    int item_set;    // which of the two item sets (work items, tile_start) is current
    // A re-bin that "rides" on a step: the step's first launch decides it, counts the tiles and builds
    // the next item set; the momentum pass and g2p of the same step then store what they store anyway
    // into the particle's slot of the OTHER store, and the last workgroup of the step commits.
    int riding;      // latched by the step's binning launch: this step's stores go to the other store
                     // (1: that launch counted the tiles, 2: the step's mass pass counts them)
    int mixing;      // latched with `riding`: the ranks of this riding re-bin were handed out to every fourth
                     // slot per wave (interleaved_thread), and the momentum pass and g2p read the same way
    int n_items[4];  // work items of the two sets, then the large ones among them (scan_tiles)
    // ---- end of the head ----
    int ticket;      // workgroups of the running binning kernel that have finished
    int age;         // g2p stages since the last re-bin
    int cum_drift;   // sum over those stages of the particles left outside their arena
    int binned;      // particles [0, binned) were there at the last re-bin (the rest arrived since)
    // Particles [0, dense) are covered by the work items.  [dense, binned) are those of the tiles the last
    // re-bin found nearly empty: a work item pays for its arena whatever it holds, a handful of particles
    // is cheaper on the follow-up kernels' path (straight to the global grid), which takes every slot from
    // `dense` on.  dense_next: of the binning a re-bin in flight is building.
    int dense, dense_next;
    int rebins;      // re-bins performed since the counter was last cleared
    // Device time of the re-bins' OWN launches, in ticks of the constant-rate wall clock: for the standalone
    // form from the first workgroup of the count launch to the commit in the scatter launch; for a riding
    // re-bin only the launch that scans the tile counts (the rest of its work is inside the step's mass
    // pass, momentum pass and g2p and is timed with them).  And the start of the one in flight.
    unsigned long long busy_ticks;
    unsigned long long stragglers;   // particles the g2p stages since then handed to the follow-up kernel
    unsigned long long t_start;      // ~0 when no re-bin is in flight
    int epoch;       // re-bins performed since the store was created (never cleared): slot indices
                     // taken under one epoch are stale under the next
    int drift_next;  // particles g2p leaves outside the arena of their NEW tile during a riding re-bin
};

// The first 32 bytes of RebinCtl as one value.
struct alignas(32) CtlHead {
    int cur, item_set, riding, mixing;
    int n_items[4];
};
static_assert(offsetof(RebinCtl, n_items) == offsetof(CtlHead, n_items) && offsetof(RebinCtl, riding) == offsetof(CtlHead, riding) &&
              offsetof(RebinCtl, item_set) == offsetof(CtlHead, item_set) && offsetof(RebinCtl, cur) == 0, "CtlHead mirrors RebinCtl");

struct TiledState {
    bool valid = false;          // false: the next decision is forced to "re-bin"
    bool checked = false;        // the decision for the current step has been enqueued
    RebinCtl *ctl = nullptr;     // device
    int tiles_x = 0, tiles_y = 0, tiles_z = 0, n_tiles = 0;
    int tiles_cap = 0;           // tiles the per-tile arrays were allocated for (>= n_tiles)
    long long item_cap = 0;
    int *key = nullptr;          // per slot: tile id
    int *rank = nullptr;         // per slot: position inside its tile's run
    int *tile_count = nullptr;   // [n_tiles]
    // Two sets of each (RebinCtl::item_set says which is current): a re-bin builds the other set
    // while the kernels of the step it rides on still read the current one.
    int *tile_start = nullptr;   // 2 x [n_tiles + 1]
    int4 *items = nullptr;       // 2 x item_cap work items {tile, begin, end, 0}
    int *n_items = nullptr;      // RebinCtl::n_items of `ctl` (no allocation of its own)
    int *straggler_counts = nullptr;   // two device scalars used by alternate stages
    int *straggler_list = nullptr;
    int chunk = kChunkMax;       // particles per work item of the current binning (choose_chunk)
    int sparse = 0;              // tiles with fewer particles get no work item (choose_sparse)
    int chunk_wanted = kChunkMax, sparse_wanted = 0;      // what the next forced re-bin takes them from
    // A riding re-bin whose tiles the coming mass pass counts: the launch that builds the next item set
    // follows that pass (launch_p2g_mass_tiled), timed with these events.
    bool scan_after_mass = false;
    int ride_interval = 0, ride_divisor = 0;      // what that mass pass decides with (rebin_due)
    hipEvent_t scan_ev_start = nullptr, scan_ev_stop = nullptr;
    int stage_parity = 0;        // which of the two the next stage counts into
    // The counter (0 / 1) under which the last momentum pass or p2g2 listed its stragglers, while the
    // particles are still the ones it saw, in the same slots; -1 otherwise.  g2p then reuses the list.
    int listed = -1;
};

// Two floats that one packed instruction (v_pk_mul_f32, v_pk_add_f32) operates on.
typedef float Float2v __attribute__((ext_vector_type(2)));

struct TileDims { int tiles_x, tiles_y, tiles_z; };

// What a tiled particle kernel needs to find its particles.
struct TiledWork {
    StoreSel store;
    const int4 *items;           // both sets (TiledState)
    const int *n_items;
    long long item_cap;
    // of the re-bin in flight (touched only while RebinCtl::riding is set): the mass pass of the step
    // writes key and rank and counts the tiles, the momentum pass and g2p read them
    int *key, *rank;
    int *tile_count;
    const int *tile_start;
    int n_tiles;
    RebinCtl *ctl;
    int n;                       // live particles: items are clipped to [0, n)
    TileDims td;
    Stragglers sg;
    // Mass pass only: particles whose floor(x) is outside [keep_lo, keep_hi) are skipped.  A slab
    // whose leavers have not been packed yet runs its mass pass this way (they are no longer its
    // particles); everywhere else the range is everything.
    int keep_lo, keep_hi;
    // A stage may be launched in two parts: part 1 takes the work items of interior tiles
    // (bnd_lo < tile x index < bnd_hi), part 2 the others; part 0 takes all.  A slab uses this to
    // run the items that cannot see its boundary planes while those planes' halo is in flight.
    int part, bnd_lo, bnd_hi;
    // Mass pass only: decide != 0 makes it the launch that decides whether a re-bin rides on the step
    // (rebin_due with these two) and latches the answer in RebinCtl::riding.
    int decide, interval, divisor;
    // Mass pass only: scan_here != 0 makes the last workgroup of the pass's follow-up kernel build the next
    // item set of a riding re-bin (scan_tiles with these two) -- a grid of few tiles needs no launch of its own
    // for that.
    int scan_here, chunk, sparse;
    // Momentum pass with the mass pass's list at hand (p2g_momentum_tiled_kernel<true>): the first
    // follow_blocks workgroups of the launch are the pass's follow-up work (straggler_p2g_team), the work items
    // start after them.  At the front they run next to the large items; at the back they were a tail of their
    // own after the last small item.  Zero in every other launch.
    int follow_blocks;
};

__device__ __forceinline__ int tile_of(float px, float py, float pz, const Params &P, TileDims td)
{
    int lx = f32_to_i32(px) - P.x_off, y = f32_to_i32(py), z = f32_to_i32(pz);
    lx = min(max(lx, 0), P.local_nx - 1);
    y = min(max(y, 0), P.gy - 1);
    z = min(max(z, 0), P.gz - 1);
    return ((lx / kTX) * td.tiles_y + (y / kTY)) * td.tiles_z + (z / kTZ);
}

// ---- binning -------------------------------------------------------------------
//
// Every step starts with two launches, bin_count_scan_kernel and bin_scatter_kernel.  Each
// workgroup of either evaluates rebin_due() first -- a pure function of control words that
// only the last workgroup of the second kernel changes -- and returns at once when no
// re-bin is due, so a step without one costs two empty launches.

// Whether this step starts with a re-bin: when the host asks for it (new particles), when the
// binning is `interval` steps old, or when the particles outside their arenas -- summed over
// the steps since the last re-bin, plus those the work items do not cover -- outweigh a
// re-bin (each such particle costs global atomics in every stage of every step; a re-bin
// costs about what n / divisor of them do).  divisor == 0 switches that rule off.
__device__ __forceinline__ bool rebin_due(const RebinCtl *ctl, int n, int interval, int force, int divisor)
{
    const int tail = n - min(ctl->binned, n);
    const long long outside = (long long)ctl->cum_drift + tail;
    return force || ctl->age >= interval || (divisor > 0 && outside * divisor > n);
}

// True in exactly one workgroup of the launch: the one that finishes last.  Everything the
// other workgroups wrote before their call is visible to it afterwards.
__device__ __forceinline__ bool last_workgroup(int *ticket)
{
    __shared__ bool last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        if (last) *ticket = 0;
    }
    __syncthreads();
    if (last) __threadfence();
    return last;
}

// Exclusive scan of the tile counts into tile_start plus the list of work items (runs of at
// most kChunk particles of one tile), by one workgroup of 256 to 1024 threads; leaves the counts zero
// for the next re-bin.  The counts were produced by device-scope atomics of other workgroups
// of the same launch, hence the device-scope loads.
//
// The list is ordered for the dispatcher, which hands out workgroups in block order: items of at least
// kLargeItem particles first, the small ones last, so that the launch ends on short workgroups (in tile
// order its last round of workgroups held full items: 550 instead of 537 us in a replay of the momentum
// pass's item durations over 1,280 slots).
//
// Tiles with fewer than `sparse` particles get no work item: their particles are placed at the end of
// the store, [n - their number, n), the others from slot 0 on, and *dense_out receives the first slot of
// that tail (RebinCtl::dense).  n: the number of particles counted.
__device__ __forceinline__ void scan_tiles(int *__restrict__ tile_count, int n_tiles, int *__restrict__ tile_start,
                                           int4 *__restrict__ items, int *__restrict__ n_items_out,
                                           long long item_cap, const int kChunk, const int n, const int sparse,
                                           int *__restrict__ dense_out)
{
    __shared__ int wave_cnt[16], wave_itm[16], wave_sml[16], wave_sp[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n_waves = (int)blockDim.x >> 6;
    int carry_cnt = 0, carry_itm = 0, carry_sml = 0, carry_sp = 0;
    for (int base = 0; base < n_tiles; base += 4 * (int)blockDim.x) {
        const int i0 = base + 4 * t;
        // per tile: particles, large work items, small work items, particles if the tile is sparse
        int c[4], m[4], ms[4], sp[4], per[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            c[q] = (i0 + q < n_tiles)
                       ? __hip_atomic_load(&tile_count[i0 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            sp[q] = c[q] < sparse ? c[q] : 0;
            c[q] -= sp[q];
            const int parts = (c[q] + kChunk - 1) / kChunk;
            // a tile that needs several items is cut into equal parts (in whole waves), not into full
            // items and a remainder: every item pays for its arena, a small one as much as a full one
            per[q] = parts > 0 ? ((c[q] + parts - 1) / parts + 63) / 64 * 64 : 0;
            const bool large = per[q] >= kLargeItem;
            m[q] = large ? parts : 0;
            ms[q] = large ? 0 : parts;
        }
        const int my_cnt = (c[0] + c[1]) + (c[2] + c[3]), my_itm = (m[0] + m[1]) + (m[2] + m[3]),
                  my_sml = (ms[0] + ms[1]) + (ms[2] + ms[3]), my_sp = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        int inc_cnt = my_cnt, inc_itm = my_itm, inc_sml = my_sml, inc_sp = my_sp;      // inclusive scans over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int a = __shfl_up(inc_cnt, off), b = __shfl_up(inc_itm, off), d = __shfl_up(inc_sp, off),
                e = __shfl_up(inc_sml, off);
            if (lane >= off) { inc_cnt += a; inc_itm += b; inc_sp += d; inc_sml += e; }
        }
        if (lane == 63) { wave_cnt[wave] = inc_cnt; wave_itm[wave] = inc_itm; wave_sml[wave] = inc_sml; wave_sp[wave] = inc_sp; }
        __syncthreads();
        int start = carry_cnt + inc_cnt - my_cnt, first_item = carry_itm + inc_itm - my_itm,
            first_small = carry_sml + inc_sml - my_sml, before_sp = carry_sp + inc_sp - my_sp;
        for (int v = 0; v < wave; v++) {
            start += wave_cnt[v]; first_item += wave_itm[v]; first_small += wave_sml[v]; before_sp += wave_sp[v];
        }
        int chunk_cnt = 0, chunk_itm = 0, chunk_sml = 0, chunk_sp = 0;
        for (int v = 0; v < n_waves; v++) {
            chunk_cnt += wave_cnt[v]; chunk_itm += wave_itm[v]; chunk_sml += wave_sml[v]; chunk_sp += wave_sp[v];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + q;
            if (i < n_tiles) {
                before_sp += sp[q];
                tile_start[i] = sp[q] ? n - before_sp : start;      // the tail is filled from its end
                tile_count[i] = 0;
                // large items from the front of the list, small ones from its back (take_item)
                for (int k = 0; k < m[q] + ms[q]; k++) {
                    const long long at = m[q] ? first_item + k : item_cap - 1 - (first_small + k);
                    if (at >= 0 && at < item_cap)
                        items[at] = make_int4(i, min(start + k * per[q], start + c[q]),
                                              min(start + (k + 1) * per[q], start + c[q]), 0);
                }
                start += c[q];
                first_item += m[q];
                first_small += ms[q];
            }
        }
        carry_cnt += chunk_cnt;
        carry_itm += chunk_itm;
        carry_sml += chunk_sml;
        carry_sp += chunk_sp;
        __syncthreads();
    }
    if (t == 0) {
        tile_start[n_tiles] = carry_cnt;
        n_items_out[0] = (int)min((long long)carry_itm + carry_sml, item_cap);
        n_items_out[2] = carry_itm;      // the large ones (the two sets' totals sit in words 0 and 1, these in 2 and 3)
        *dense_out = carry_cnt;
    }
}

// First launch: tile id per particle and its arrival rank inside the tile, then (last
// workgroup) the scan.  Lanes of a wave that hold the same tile share ONE atomic on the tile counter.
__global__ void __launch_bounds__(256)
bin_count_scan_kernel(StoreSel ss, int n, Params P, TileDims td, int *__restrict__ key, int *__restrict__ rank,
                      int *__restrict__ tile_count, int n_tiles, int *__restrict__ tile_start,
                      int4 *__restrict__ items, int *__restrict__ n_items_out, long long item_cap,
                      RebinCtl *ctl, int interval, int force, int divisor, int ride, int chunk, int sparse)
{
    // tile_start, items, n_items_out: both sets; this launch fills the one that is not current.
    // ride: the re-bin, if due, rides on this step (RebinCtl::riding) instead of being completed by
    // bin_scatter_kernel.  The decision is a pure function of words nothing changes before g2p.
    const bool due = rebin_due(ctl, n, interval, force, divisor);
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->riding = (ride && due) ? 1 : 0; ctl->mixing = 0; }
    if (!due) return;
    const int next_set = ctl->item_set ^ 1;
    tile_start += (long long)next_set * (n_tiles + 1);
    items += (long long)next_set * item_cap;
    n_items_out += next_set;
    if (threadIdx.x == 0) atomicMin(&ctl->t_start, (unsigned long long)wall_clock64());
    const Store st = ss.pick();
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    // Each workgroup takes one contiguous run of particles: the store is nearly sorted, so a run
    // touches two or three tile counters and few other workgroups touch the same ones (with a
    // grid-stride loop every counter would be contended by dozens of workgroups at a time).
    const long long per = (((long long)n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long run_begin = (long long)blockIdx.x * per, run_end = min((long long)n, run_begin + per);
    for (long long j0 = run_begin; j0 < run_end; j0 += blockDim.x) {
        const long long j = j0 + threadIdx.x;
        const bool valid = j < run_end;
        int k = -1 - lane;   // distinct for idle lanes, never equal to a tile id
        if (valid) k = tile_of(st.slot(kPosX)[j], st.slot(kPosY)[j], st.slot(kPosZ)[j], P, td);
        // One atomic per distinct tile among the 64 particles of the wave, wherever its lanes are: by
        // the time a re-bin is due a quarter of the particles sit in a tile next to the one they were
        // binned under, so runs of equal keys are short, and one atomic per run was 0.45 ms against
        // 0.18 ms this way (tools/ubench/count_pass.hip).  The groups are found first, without
        // memory operations; then all their leaders issue their atomics at once.
        int leader = lane, pos = 0, size = 0;
        for (unsigned long long rest = __ballot(valid); rest != 0;) {          // uniform
            const int l = __ffsll((long long)rest) - 1;
            const int kl = __shfl(k, l);
            const unsigned long long same = __ballot(k == kl);                  // idle lanes hold unique negative keys
            if (k == kl) { leader = l; pos = __popcll(same & below); size = __popcll(same); }
            rest &= ~same;
        }
        int base = 0;
        if (valid && lane == leader) base = atomicAdd(&tile_count[k], size);
        base = __shfl(base, leader);
        if (valid) { key[j] = k; rank[j] = base + pos; }
    }
    if (last_workgroup(&ctl->ticket)) {
        scan_tiles(tile_count, n_tiles, tile_start, items, n_items_out, item_cap, chunk, n, sparse, &ctl->dense_next);
        if (ride && threadIdx.x == 0) {      // a riding re-bin's own device time ends here
            ctl->busy_ticks += (unsigned long long)wall_clock64() - ctl->t_start;
            ctl->t_start = ~0ULL;
        }
    }
}

// Second launch: every record moves to its slot of the other store (reads coalesced; writes
// nearly so, because the store was binned a few steps ago), then (last workgroup) the other
// store becomes current.
__global__ void __launch_bounds__(256)
bin_scatter_kernel(StoreSel ss, int n, const int *__restrict__ key, const int *__restrict__ rank,
                   const int *__restrict__ tile_start, int n_tiles, RebinCtl *ctl, int interval, int force,
                   int divisor)
{
    if (!rebin_due(ctl, n, interval, force, divisor)) return;
    tile_start += (long long)(ctl->item_set ^ 1) * (n_tiles + 1);      // the set the first launch filled
    const Store src = ss.pick(), dst = ss.other();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const int d = tile_start[key[j]] + rank[j];
#pragma unroll
        for (int w = 0; w < kRecordWords; w++) dst.slot(w)[d] = src.slot(w)[j];
    }
    if (last_workgroup(&ctl->ticket) && threadIdx.x == 0) {
        ctl->cur ^= 1;
        ctl->item_set ^= 1;
        ctl->age = 0;
        ctl->cum_drift = 0;
        ctl->binned = n;
        ctl->dense = ctl->dense_next;
        ctl->rebins += 1;
        ctl->epoch += 1;
        ctl->busy_ticks += (unsigned long long)wall_clock64() - ctl->t_start;
        ctl->t_start = ~0ULL;
    }
}

// A riding re-bin needs no counting launch of its own: the mass pass of the step reads every position
// anyway, and a work item's particles have moved, if at all, into one of the 26 tiles around their own --
// so a workgroup counts them in 27 LDS words and takes its share of the global tile counters with at
// most 27 atomics.  The mass pass itself decides (every workgroup evaluates rebin_due, a pure function of
// words nothing changes before g2p; its first workgroup latches the answer for the launches that follow),
// ride_scan_kernel (one workgroup) turns the counts into the next item set after it.
__global__ void __launch_bounds__(1024)
ride_scan_kernel(RebinCtl *ctl, int *__restrict__ tile_count, int n_tiles, int *__restrict__ tile_start,
                 int4 *__restrict__ items, int *__restrict__ n_items_out, long long item_cap, int chunk, int n,
                 int sparse)
{
    if (ctl->riding != 2) return;
    const unsigned long long t0 = wall_clock64();
    const int next_set = ctl->item_set ^ 1;
    scan_tiles(tile_count, n_tiles, tile_start + (long long)next_set * (n_tiles + 1), items + (long long)next_set * item_cap,
               n_items_out + next_set, item_cap, chunk, n, sparse, &ctl->dense_next);
    if (threadIdx.x == 0) ctl->busy_ticks += (unsigned long long)wall_clock64() - t0;
}

// rank[] entries between the two passes of a counting mass pass
constexpr int kRankNotMine = 0x7FFFFFFF;       // a straggler: the follow-up kernel ranks it
constexpr int kRankNbShift = 24;               // (neighbour index << 24) | rank inside this workgroup's count

// ---- arena helpers ---------------------------------------------------------------

struct Arena {
    int ox, oy, oz;      // global cell coordinates of arena cell (0,0,0)
};

__device__ __forceinline__ Arena arena_of_tile(int tile, TileDims td, const Params &P)
{
    int tz = tile % td.tiles_z;
    int ty = (tile / td.tiles_z) % td.tiles_y;
    int tx = tile / (td.tiles_z * td.tiles_y);
    return Arena{tx * kTX + P.x_off - kRim, ty * kTY - kRim, tz * kTZ - kRim};
}

__device__ __forceinline__ int arena_index(int ax, int ay, int az) { return (ax * kAY + ay) * kAZ + az; }

// Global (local-grid) linear index of arena cell a, or -1 outside the held planes.
__device__ __forceinline__ long long arena_cell_to_grid(int a, const Arena &A, const Params &P)
{
    int az = a % kAZ, ay = (a / kAZ) % kAY, ax = a / (kAZ * kAY);
    return local_cell(A.ox + ax, A.oy + ay, A.oz + az, P);
}

// A workgroup collects its stragglers in LDS and appends them to the list with ONE atomic on the
// shared counter (an atomic per straggler, all on that one word and each waiting for its return
// value, cost the tiled kernels 5-6 ns per straggler); what does not fit goes there directly.
constexpr int kStragglerBuf = 126;
struct StragglerBuf { int n, base, slot[kStragglerBuf]; };

__device__ __forceinline__ void add_straggler_direct(const Stragglers &sg, int j)
{
    sg.list[atomicAdd(sg.count, 1)] = j;
}

// before the first barrier of the kernel
__device__ __forceinline__ void straggler_buf_init(StragglerBuf &b) { if (threadIdx.x == 0) b.n = 0; }

__device__ __forceinline__ void add_straggler(const Stragglers &sg, StragglerBuf &b, int j)
{
    const int k = atomicAdd(&b.n, 1);
    if (k < kStragglerBuf) b.slot[k] = j;
    else sg.list[atomicAdd(sg.count, 1)] = j;
}

// after the particle loop, by every thread of the workgroup; starts with a barrier
__device__ __forceinline__ void straggler_buf_flush(const Stragglers &sg, StragglerBuf &b)
{
    __syncthreads();
    const int n = min(b.n, kStragglerBuf);
    if (n == 0) return;                                 // uniform
    if (threadIdx.x == 0) b.base = atomicAdd(sg.count, n);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) sg.list[b.base + k] = b.slot[k];
}

// Adds every non-zero word of an LDS arena component to the global grid.  A wave
// instruction covers (component, consecutive cells) so that its 64 atomics fall
// in runs of contiguous 16-byte cell records.
// Which of the 27 tiles an arena overlaps the 3x3x3 stencil with arena base (bx, by, bz) reaches:
// bit (ox * 3 + oy) * 3 + oz with o = 0, 1, 2 for the tile before, the arena's own tile, the tile
// after.  Per axis the stencil spans cells b .. b + 2, hence one or two of the three.
__device__ __forceinline__ unsigned stencil_tile_bits(int bx, int by, int bz)
{
    auto axis = [](int b, int tile_size) -> unsigned {          // 3-bit set of tile offsets
        const int lo = b < kRim ? 0 : (b < kRim + tile_size ? 1 : 2);
        const int hi = b + 2 < kRim ? 0 : (b + 2 < kRim + tile_size ? 1 : 2);
        return (1u << lo) | (1u << hi);
    };
    auto spread = [](unsigned m, int by_bits) -> unsigned {     // bit k of m -> bit k * by_bits
        return (m & 1u) | ((m & 2u) << (by_bits - 1)) | ((m & 4u) << (2 * by_bits - 2));
    };
    const unsigned z = axis(bz, kTZ);
    const unsigned yz = z * spread(axis(by, kTY), 3);           // no carries: the factors' bits do not overlap
    return yz * spread(axis(bx, kTX), 9);
}

// Sets bits in a tile's word.  Many particles mark the same few tiles, and atomics on one word are
// served one after the other, so the word is read first: within a kernel bits are only ever set, and a
// bit that reads as set needs nothing.
__device__ __forceinline__ void set_dirty(int *word, int bits)
{
    if ((*word & bits) != bits) atomicOr(word, bits);
}

// `touched` holds, per thread, the stencil_tile_bits of every particle it scattered into the
// arena of `tile`.  Sets g.dirty_bits on those tiles.  All threads of the workgroup call it.
__device__ __forceinline__ void mark_dirty_tiles(unsigned touched, int tile, TileDims td, const FixedGrid &g)
{
    if (!g.dirty) return;                   // uniform
    __shared__ unsigned touched_all;
    if (threadIdx.x == 0) touched_all = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) touched |= __shfl_xor(touched, off);
    if ((threadIdx.x & 63) == 0 && touched) atomicOr(&touched_all, touched);
    __syncthreads();
    if (threadIdx.x < 27 && (touched_all >> threadIdx.x & 1u)) {
        const int tz = tile % td.tiles_z, ty = (tile / td.tiles_z) % td.tiles_y, tx = tile / (td.tiles_z * td.tiles_y);
        const int k = threadIdx.x;
        const int nx = tx + k / 9 - 1, ny = ty + (k / 3) % 3 - 1, nz = tz + k % 3 - 1;
        // a cell that exists lies in a tile that exists
        set_dirty(&g.dirty[(nx * td.tiles_y + ny) * td.tiles_z + nz], g.dirty_bits);
    }
}

// Sets g.dirty_bits on the (up to 8) tiles the 27 cells of a stencil lie in.
__device__ __forceinline__ void mark_dirty_stencil(const Stencil &s, TileDims td, const Params &P, const FixedGrid &g)
{
    if (!g.dirty) return;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int cx = s.cx + (k & 4 ? 2 : 0), cy = s.cy + (k & 2 ? 2 : 0), cz = s.cz + (k & 1 ? 2 : 0);
        const int lx = cx - P.x_off;
        if ((unsigned)lx >= (unsigned)P.local_nx || (unsigned)cy >= (unsigned)P.gy || (unsigned)cz >= (unsigned)P.gz)
            continue;
        set_dirty(&g.dirty[((lx / kTX) * td.tiles_y + cy / kTY) * td.tiles_z + cz / kTZ], g.dirty_bits);
    }
}

template <int NCOMP>
__device__ __forceinline__ void flush_arena(const int *lds, const Arena &A, const Params &P,
                                            FixedGrid g)
{
    for (int f = threadIdx.x; f < kArenaCells * 4; f += kTiledThreads) {
        int comp = f & 3, a = f >> 2;
        if (comp >= NCOMP) continue;
        int v = lds[comp * kArenaPad + a];
        if (v == 0) continue;
        long long c = arena_cell_to_grid(a, A, P);
        if (c < 0) continue;
        if (comp == 3) atomicAdd(g.mass + c, v);
        else atomicAdd(g.cell + 4 * c + comp, v);
    }
}

// The work item of a workgroup, or -1 for the surplus blocks of the launch bound.  Workgroups are dealt
// round-robin over the 8 XCDs, each with its own L2; handing every XCD one contiguous run of items
// (neighbouring tiles share arena cells) measured 3 % slower on the ellipsoid scene, because such runs
// hold unequal work, so block b simply takes item b.
__device__ __forceinline__ int item_for_block(int n_items, int first_block)
{
    const int b = (int)blockIdx.x - first_block;
    if (b >= n_items) return -1;
    return b;
}

// Which slot of a trip's kTiledThreads consecutive slots a thread takes when a wave is to hold every fourth slot
// instead of 64 consecutive ones: lane l of wave w takes slot l * (waves per workgroup) + w.  The mass pass reads
// this way (except when it counts for a re-bin that keeps the order); the momentum pass and g2p only on the step
// of a re-bin whose ranks were handed out this way (RebinCtl::mixing), so that their stores are contiguous.
// Particles that sit next to each other in the store tend to sit in the same cell once the fluid has
// moved for a while (a re-bin keeps the order of those that stay and appends those that arrive together; at
// 2^24 particles / 256^3, 220 steps in, two neighbouring slots share a cell with probability 0.10, slots four
// apart with 0.03, random ones with 0.002 -- tools/slot_coherence.py), and lanes of a wave that share a cell
// add to the same 27 or 81 LDS words one after the other.  The workgroup's waves still read and write the same
// cache lines at the same time, a quarter of each.
__device__ __forceinline__ int interleaved_thread()
{
    constexpr int kWaves = kTiledThreads / 64;
    return (int)(threadIdx.x & 63u) * kWaves + (int)(threadIdx.x >> 6);
}

// The head of the control words, by one scalar load.
__device__ __forceinline__ CtlHead load_head(const RebinCtl *ctl) { return *reinterpret_cast<const CtlHead *>(ctl); }

// The work item of this workgroup, clipped to the live particles; false when there is none.
__device__ __forceinline__ bool take_item(const TiledWork &w, const CtlHead &h, int4 *item)
{
    const int set = h.item_set;
    // (selects, not h.n_items[set]: a dynamically indexed copy of the head would live in scratch memory)
    const int id = item_for_block(set ? h.n_items[1] : h.n_items[0], w.follow_blocks);
    if (id < 0) return false;
    const int large = set ? h.n_items[3] : h.n_items[2];      // these from the front of the list, the small ones from its back
    *item = w.items[(long long)set * w.item_cap + (id < large ? id : w.item_cap - 1 - (id - large))];
    item->z = min(item->z, w.n);
    if (w.part != 0) {
        const int tx = item->x / (w.td.tiles_y * w.td.tiles_z);
        const bool boundary = tx <= w.bnd_lo || tx >= w.bnd_hi;
        if (boundary != (w.part == 2)) return false;
    }
    return item->y < item->z;
}
__device__ __forceinline__ bool take_item(const TiledWork &w, int4 *item) { return take_item(w, load_head(w.ctl), item); }

// Where a kernel stores a particle's record while a re-bin rides on the step: slot
// tile_start[key] + rank of the other store (key, rank and the next set's tile_start were written by
// this step's binning launch).  Off: the particle's own slot of the current store.
struct Ride {
    bool on;
    Store dst;
    const int *key, *rank, *start;
    __device__ __forceinline__ int slot(int j) const { return on ? start[key[j]] + rank[j] : j; }
};

// The columns no stage of the step writes, from slot j of the current store to slot d of the other.
__device__ __forceinline__ void ride_carry_constants(const Store &src, const Store &dst, int j, int d)
{
    dst.slot(kMass)[d] = src.slot(kMass)[j];
    dst.slot(kColR)[d] = src.slot(kColR)[j];
    dst.slot(kColG)[d] = src.slot(kColG)[j];
    dst.slot(kColB)[d] = src.slot(kColB)[j];
    dst.ids()[d] = src.ids()[j];
}

__device__ __forceinline__ Ride ride_of(const TiledWork &w, const CtlHead &h, const Store &cur)
{
    const bool on = h.riding != 0;
    const Store other{h.cur ? w.store.base[0] : w.store.base[1], w.store.capacity};
    return Ride{on, on ? other : cur, w.key, w.rank, w.tile_start + (long long)(h.item_set ^ 1) * (w.n_tiles + 1)};
}
__device__ __forceinline__ Ride ride_of(const TiledWork &w, const Store &cur) { return ride_of(w, load_head(w.ctl), cur); }

// ---- p2g1 (SIM:295-333) ------------------------------------------------------------

// One particle's 27 x 4 contributions.  INSIDE: the whole stencil lies in the
// arena and goes to LDS; otherwise (a particle that drifted out of its bin since
// the last re-bin) it goes straight to the global grid.
template <bool INSIDE>
__device__ __forceinline__ void p2g1_particle(const Stencil &s, float vx, float vy, float vz, const float C[9],
                                              int *lds, int abase, FixedGrid g, const Params &P)
{
#pragma unroll
    for (int gx = 0; gx < 3; gx++) {
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                float dx = s.dx[gx], dy = s.dy[gy], dz = s.dz[gz];
                float qx = (C[0] * dx + C[3] * dy) + C[6] * dz;
                float qy = (C[1] * dx + C[4] * dy) + C[7] * dz;
                float qz = (C[2] * dx + C[5] * dy) + C[8] * dz;
                int ex = encode_fixed(weight * (vx + qx));
                int ey = encode_fixed(weight * (vy + qy));
                int ez = encode_fixed(weight * (vz + qz));
                int em = encode_fixed(weight);
                if (INSIDE) {
                    int a = abase + arena_index(gx, gy, gz);
                    atomicAdd(&lds[0 * kArenaPad + a], ex);
                    atomicAdd(&lds[1 * kArenaPad + a], ey);
                    atomicAdd(&lds[2 * kArenaPad + a], ez);
                    atomicAdd(&lds[3 * kArenaPad + a], em);
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    if (c >= 0) {
                        atomicAdd(g.cell + 4 * c + 0, ex);
                        atomicAdd(g.cell + 4 * c + 1, ey);
                        atomicAdd(g.cell + 4 * c + 2, ez);
                        atomicAdd(g.mass + c, em);
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ bool stencil_in_arena(const Stencil &s, const Arena &A, int *abase)
{
    const int bx = s.cx - A.ox, by = s.cy - A.oy, bz = s.cz - A.oz;   // arena coords of the stencil base
    *abase = arena_index(bx, by, bz);
    return (unsigned)bx <= (unsigned)(kAX - 3) && (unsigned)by <= (unsigned)(kAY - 3) &&
           (unsigned)bz <= (unsigned)(kAZ - 3);
}

__global__ void __launch_bounds__(kTiledThreads)
p2g1_tiled_kernel(TiledWork w, FixedGrid g, Params P)
{
    const Store st = w.store.pick();
    int4 item;
    if (!take_item(w, &item)) return;
    const Stragglers sg = w.sg;
    __shared__ int lds[4 * kArenaPad];
    __shared__ StragglerBuf sb;
    straggler_buf_init(sb);
    const Arena A = arena_of_tile(item.x, w.td, P);
    for (int i = threadIdx.x; i < 4 * kArenaPad; i += kTiledThreads) lds[i] = 0;
    __syncthreads();

    unsigned touched = 0;      // see p2g_mass_tiled_kernel; p2g2 covers the same cells
    for (int j = item.y + threadIdx.x; j < item.z; j += kTiledThreads) {
        float px = st.slot(kPosX)[j], py = st.slot(kPosY)[j], pz = st.slot(kPosZ)[j];
        float vx = st.slot(kVelX)[j], vy = st.slot(kVelY)[j], vz = st.slot(kVelZ)[j];
        float C[9];
#pragma unroll
        for (int k = 0; k < 9; k++) C[k] = st.slot(kC00 + k)[j];
        Stencil s = make_stencil(px, py, pz);
        int abase;
        if (stencil_in_arena(s, A, &abase)) {
            p2g1_particle<true>(s, vx, vy, vz, C, lds, abase, g, P);
            touched |= stencil_tile_bits(s.cx - A.ox, s.cy - A.oy, s.cz - A.oz);
        } else
            add_straggler(sg, sb, j);
    }
    straggler_buf_flush(sg, sb);
    __syncthreads();
    flush_arena<4>(lds, A, P, g);
    mark_dirty_tiles(touched, item.x, w.td, g);
}

// ---- p2g2 (SIM:336-392) ------------------------------------------------------------

template <bool INSIDE>
__device__ __forceinline__ void p2g2_particle(const Stencil &s, const float C[9], float *dens_slot, int *lds,
                                              const float *mass_f, int abase, FixedGrid g,
                                              const Params &P)
{
    float density = 0.0f;
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                float m;
                if (INSIDE) {
                    m = mass_f[abase + arena_index(gx, gy, gz)];
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    m = decode_fixed((c >= 0) ? g.mass[c] : 0);
                }
                density = density + m * weight;
            }
        }
    *dens_slot = *dens_slot * (1.0f - 0.05f) + density * 0.05f;      // SIM:365-366
    Eq16 e = p2g2_stress(density, C, P);
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                float dx = s.dx[gx], dy = s.dy[gy], dz = s.dz[gz];
                float a00 = e.m00 * weight, a11 = e.m11 * weight, a22 = e.m22 * weight;
                float a01 = e.m01 * weight, a02 = e.m02 * weight, a12 = e.m12 * weight;
                int ex = encode_fixed((a00 * dx + a01 * dy) + a02 * dz);
                int ey = encode_fixed((a01 * dx + a11 * dy) + a12 * dz);
                int ez = encode_fixed((a02 * dx + a12 * dy) + a22 * dz);
                if (INSIDE) {
                    int a = abase + arena_index(gx, gy, gz);
                    atomicAdd(&lds[0 * kArenaPad + a], ex);
                    atomicAdd(&lds[1 * kArenaPad + a], ey);
                    atomicAdd(&lds[2 * kArenaPad + a], ez);
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    if (c >= 0) {
                        atomicAdd(g.cell + 4 * c + 0, ex);
                        atomicAdd(g.cell + 4 * c + 1, ey);
                        atomicAdd(g.cell + 4 * c + 2, ez);
                    }
                }
            }
        }
}

__global__ void __launch_bounds__(kTiledThreads)
p2g2_tiled_kernel(TiledWork w, FixedGrid g, Params P)
{
    const Store st = w.store.pick();
    int4 item;
    if (!take_item(w, &item)) return;
    const Stragglers sg = w.sg;
    __shared__ int lds[3 * kArenaPad];
    __shared__ float mass_f[kArenaCells];      // decodeFixedPoint(cell.mass), once per arena cell
    __shared__ StragglerBuf sb;
    straggler_buf_init(sb);
    const Arena A = arena_of_tile(item.x, w.td, P);
    for (int i = threadIdx.x; i < 3 * kArenaPad; i += kTiledThreads) lds[i] = 0;
    for (int a = threadIdx.x; a < kArenaCells; a += kTiledThreads) {
        long long c = arena_cell_to_grid(a, A, P);
        int m = (c >= 0) ? g.mass[c] : 0;
        mass_f[a] = decode_fixed(m);
    }
    __syncthreads();

    const Column<float> dens = st.slot(kDensity);
    for (int j = item.y + threadIdx.x; j < item.z; j += kTiledThreads) {
        float px = st.slot(kPosX)[j], py = st.slot(kPosY)[j], pz = st.slot(kPosZ)[j];
        float C[9];
#pragma unroll
        for (int k = 0; k < 9; k++) C[k] = st.slot(kC00 + k)[j];
        Stencil s = make_stencil(px, py, pz);
        int abase;
        if (stencil_in_arena(s, A, &abase))
            p2g2_particle<true>(s, C, dens + j, lds, mass_f, abase, g, P);
        else
            add_straggler(sg, sb, j);
    }
    straggler_buf_flush(sg, sb);
    __syncthreads();
    flush_arena<3>(lds, A, P, g);
}

// ---- fused p2g: mass pass + momentum pass --------------------------------------------
//
// p2g1 and p2g2 together perform 27 x (4 + 3) atomic adds per particle.  The momentum
// words of p2g1 are not read by anything before updateGrid, and integer addition is
// associative, so the step is free to add encode(p2g1 momentum) and encode(p2g2 momentum)
// of one (particle, cell) pair in a register and issue ONE atomic for the sum -- as long
// as each term is still encoded on its own first (SIM:279-281, 326-328, 386-388).  That
// gives a mass-only first pass (27 adds, reads position only) and a second pass with
// 27 x 3 adds: 108 instead of 189 LDS atomics per particle, and the grid after the second
// pass is bit-identical to the grid after the reference's p2g2.  The grid between the two
// passes differs from the reference's (momentum not yet present), so flowmpm_run_kernel
// keeps the unfused kernels above for stage-by-stage parity.

__device__ __forceinline__ int add_wrap(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

template <bool INSIDE>
__device__ __forceinline__ void p2g_mass_particle(const Stencil &s, int *lds, int abase, FixedGrid g,
                                                  const Params &P)
{
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                int em = encode_fixed(wxy * s.wz[gz]);
                if (INSIDE) {
                    atomicAdd(&lds[abase + arena_index(gx, gy, gz)], em);
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    if (c >= 0) atomicAdd(g.mass + c, em);
                }
            }
        }
}

__global__ void __launch_bounds__(kTiledThreads)
p2g_mass_tiled_kernel(TiledWork w, FixedGrid g, Params P)
{
    const CtlHead head = load_head(w.ctl);
    const Store st{head.cur ? w.store.base[1] : w.store.base[0], w.store.capacity};
    // counting for a riding re-bin (uniform over the launch)
    const bool counting = w.decide ? rebin_due(w.ctl, w.n, w.interval, 0, w.divisor) : head.riding == 2;
    // Every kMixEvery-th riding re-bin mixes the store order (see the end of this kernel): a mixing re-bin makes
    // the momentum pass and g2p of its step read with a stride, which costs them 0.3 ms each at 2^24 particles.
    const bool mixing = counting && (w.decide ? (w.ctl->epoch % kMixEvery) == 0 : head.mixing != 0);
    if (w.decide && blockIdx.x == 0 && threadIdx.x == 0) {
        w.ctl->riding = counting ? 2 : 0;
        w.ctl->mixing = mixing ? 1 : 0;
    }
    int4 item;
    if (!take_item(w, head, &item)) return;
    const Stragglers sg = w.sg;
    __shared__ int lds[kArenaCells];
    __shared__ StragglerBuf sb;
    __shared__ int nb_count[27], nb_base[27];
    straggler_buf_init(sb);
    const Arena A = arena_of_tile(item.x, w.td, P);
    // The first positions are requested before the arena is zeroed, and every later trip requests the next
    // ones before it issues its atomics: a work item of this pass is a chain of memory round trips (the
    // compute units hold all the workgroups they can, eight each), and each of them taken off the chain
    // shortens the pass.
    // (a counting pass that does not mix keeps the slots in order: the ranks it hands out decide where the
    // momentum pass and g2p of the step, which then read their particles in order, store them)
    int j = item.y + (counting && !mixing ? (int)threadIdx.x : interleaved_thread());
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (j < item.z) { px = st.slot(kPosX)[j]; py = st.slot(kPosY)[j]; pz = st.slot(kPosZ)[j]; }
    for (int i = threadIdx.x; i < kArenaCells; i += kTiledThreads) lds[i] = 0;
    const int tz0 = item.x % w.td.tiles_z, ty0 = (item.x / w.td.tiles_z) % w.td.tiles_y,
              tx0 = item.x / (w.td.tiles_z * w.td.tiles_y);
    if (threadIdx.x < 27) nb_count[threadIdx.x] = 0;
    __syncthreads();
    // The tiles this work item adds to, for clearGrid and updateGrid (FixedGrid::dirty).  The
    // momentum pass scatters the same particles over the same cells and marks nothing.
    unsigned touched = 0;
    for (int jn; j < item.z; j = jn) {
        Stencil s = make_stencil(px, py, pz);
        jn = j + kTiledThreads;
        if (jn < item.z) { px = st.slot(kPosX)[jn]; py = st.slot(kPosY)[jn]; pz = st.slot(kPosZ)[jn]; }
        // a slab's leavers (floor(x) outside the planes it owns) are about to be packed away: they are neither
        // scattered nor counted
        const bool leaver = s.cx + 1 < w.keep_lo || s.cx + 1 >= w.keep_hi;       // s.cx = floor(x) - 1
        if (counting) {
            int abase_unused;
            if (leaver || !stencil_in_arena(s, A, &abase_unused)) {
                w.rank[j] = kRankNotMine;
            } else {
                // tile of floor(position) = s.c + 1; inside the arena it is this tile or one next to it
                // (clamped to the grid like tile_of: a position on the top face of a grid whose extent is a
                // multiple of the tile size would otherwise name a tile that does not exist)
                const int kx = min((s.cx + 1 - P.x_off) / kTX, w.td.tiles_x - 1), ky = min((s.cy + 1) / kTY, w.td.tiles_y - 1),
                          kz = min((s.cz + 1) / kTZ, w.td.tiles_z - 1);
                const int nb = ((kx - tx0 + 1) * 3 + (ky - ty0 + 1)) * 3 + (kz - tz0 + 1);
                w.key[j] = (kx * w.td.tiles_y + ky) * w.td.tiles_z + kz;
                w.rank[j] = (nb << kRankNbShift) | atomicAdd(&nb_count[nb], 1);
            }
        }
        if (leaver) continue;
        int abase;
        if (stencil_in_arena(s, A, &abase)) {
            p2g_mass_particle<true>(s, lds, abase, g, P);
            touched |= stencil_tile_bits(s.cx - A.ox, s.cy - A.oy, s.cz - A.oz);
        } else
            add_straggler(sg, sb, j);
    }
    straggler_buf_flush(sg, sb);
    __syncthreads();
    for (int a = threadIdx.x; a < kArenaCells; a += kTiledThreads) {
        int v = lds[a];
        if (v == 0) continue;
        long long c = arena_cell_to_grid(a, A, P);
        if (c >= 0) atomicAdd(g.mass + c, v);
    }
    mark_dirty_tiles(touched, item.x, w.td, g);
    if (!counting) return;                                   // uniform
    // this workgroup's share of the 27 tile counters, then the ranks become ranks inside the tile
    if (threadIdx.x < 27 && nb_count[threadIdx.x] > 0) {
        const int nb = threadIdx.x;
        const int tile = ((tx0 + nb / 9 - 1) * w.td.tiles_y + (ty0 + (nb / 3) % 3 - 1)) * w.td.tiles_z + (tz0 + nb % 3 - 1);
        nb_base[nb] = atomicAdd(&w.tile_count[tile], nb_count[nb]);
    }
    __syncthreads();
    // Inside the block this workgroup reserved in a tile, whole runs of 64 ranks -- what one wave of this pass
    // counted together, and what one wave of the momentum pass and of g2p will store together -- change places
    // (run q goes where run q * p mod runs was, p coprime with the number of runs): the interleaving of the
    // lanes alone only ever mixes the 256 slots of one trip among themselves, and four re-bins in a row put
    // them back in their old order.
    for (int j = item.y + threadIdx.x; j < item.z; j += kTiledThreads) {
        const int r = w.rank[j];
        if (r == kRankNotMine) continue;
        const int nb = r >> kRankNbShift;
        int local = r & ((1 << kRankNbShift) - 1);
        const unsigned runs = (unsigned)nb_count[nb] >> 6;
        const unsigned run = (unsigned)local >> 6;
        if (run < runs && runs > 2u) {
            const unsigned p = (runs % 257u) ? 257u : 263u;
            local = (int)(((run * p) % runs) << 6) | (local & 63);
        }
        w.rank[j] = nb_base[nb] + local;
    }
}

template <bool INSIDE>
__device__ __forceinline__ void p2g_momentum_particle(const Stencil &s, float vx, float vy, float vz,
                                                      const float C[9], const float *dens_slot, float *dens_out,
                                                      int *lds, const float *mass_f, int abase, FixedGrid g,
                                                      const Params &P)
{
    float density = 0.0f;
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                float m;
                if (INSIDE) {
                    m = mass_f[abase + arena_index(gx, gy, gz)];
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    m = decode_fixed((c >= 0) ? g.mass[c] : 0);
                }
                density = density + m * weight;
            }
        }
    *dens_out = *dens_slot * (1.0f - 0.05f) + density * 0.05f;      // SIM:365-366
    Eq16 e = p2g2_stress(density, C, P);
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                float dx = s.dx[gx], dy = s.dy[gy], dz = s.dz[gz];
                // p2g1's velContrib (SIM:321-328)
                float qx = (C[0] * dx + C[3] * dy) + C[6] * dz;
                float qy = (C[1] * dx + C[4] * dy) + C[7] * dz;
                float qz = (C[2] * dx + C[5] * dy) + C[8] * dz;
                int ex = encode_fixed(weight * (vx + qx));
                int ey = encode_fixed(weight * (vy + qy));
                int ez = encode_fixed(weight * (vz + qz));
                // p2g2's momentum (SIM:385-388)
                float a00 = e.m00 * weight, a11 = e.m11 * weight, a22 = e.m22 * weight;
                float a01 = e.m01 * weight, a02 = e.m02 * weight, a12 = e.m12 * weight;
                ex = add_wrap(ex, encode_fixed((a00 * dx + a01 * dy) + a02 * dz));
                ey = add_wrap(ey, encode_fixed((a01 * dx + a11 * dy) + a12 * dz));
                ez = add_wrap(ez, encode_fixed((a02 * dx + a12 * dy) + a22 * dz));
                if (INSIDE) {
                    int a = abase + arena_index(gx, gy, gz);
                    atomicAdd(&lds[0 * kArenaPad + a], ex);
                    atomicAdd(&lds[1 * kArenaPad + a], ey);
                    atomicAdd(&lds[2 * kArenaPad + a], ez);
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    if (c >= 0) {
                        atomicAdd(g.cell + 4 * c + 0, ex);
                        atomicAdd(g.cell + 4 * c + 1, ey);
                        atomicAdd(g.cell + 4 * c + 2, ez);
                    }
                }
            }
        }
}

template <bool MOMENTUM>
__device__ __forceinline__ void straggler_p2g_team(const TiledWork &w, const FixedGrid &g, const Params &P, int block, int blocks);

// While a re-bin rides on the step, the momentum pass (bound by LDS and VALU, with memory bandwidth to
// spare) also carries the columns no stage writes -- mass, colour, id -- to the particle's new slot.
//
// LISTED: the mass pass of this step ran on the same particles in the same slots and left the list of those
// whose stencil is outside their arena (TiledState::listed).  This pass then lists nothing, and its follow-up
// work -- those particles, straight to the global grid -- is done by the first workgroups of the same launch
// (TiledWork::follow_blocks) instead of a launch of its own.
template <bool LISTED>
__global__ void __launch_bounds__(kTiledThreads, FLOWMPM_P2G_WAVES)
p2g_momentum_tiled_kernel(TiledWork w, FixedGrid g, Params P)
{
    if (LISTED && (int)blockIdx.x < w.follow_blocks) {      // uniform
        straggler_p2g_team<true>(w, g, P, (int)blockIdx.x, w.follow_blocks);
        return;
    }
    const CtlHead head = load_head(w.ctl);
    const Store st{head.cur ? w.store.base[1] : w.store.base[0], w.store.capacity};
    int4 item;
    if (!take_item(w, head, &item)) return;
    const Stragglers sg = w.sg;
    ClockStamp stamp;
    stamp.entered();
    __shared__ int lds[3 * kArenaPad];
    __shared__ float mass_f[kArenaCells];
    __shared__ StragglerBuf sb;
    straggler_buf_init(sb);
    const Arena A = arena_of_tile(item.x, w.td, P);
    // (Requesting the first particles before or together with the arena, which gains g2p 4 %, was
    // measured here too: no change when it costs no registers, and a wave per SIMD less when it does.)
    for (int i = threadIdx.x; i < 3 * kArenaPad; i += kTiledThreads) lds[i] = 0;
    for (int a = threadIdx.x; a < kArenaCells; a += kTiledThreads) {
        long long c = arena_cell_to_grid(a, A, P);
        mass_f[a] = decode_fixed((c >= 0) ? g.mass[c] : 0);
    }
    __syncthreads();
    const Column<float> dens = st.slot(kDensity);
    const Ride ride = ride_of(w, head, st);
    const Column<float> dens_out = ride.dst.slot(kDensity);
    float in[15];
    // (on the step of a mixing re-bin: the mapping under which the mass pass ranked the particles, so that a
    // wave's lanes store to consecutive slots of the new binning)
    int j = item.y + (head.riding != 0 && head.mixing != 0 ? interleaved_thread() : (int)threadIdx.x);
    if (j < item.z) {
#pragma unroll
        for (int k = 0; k < 15; k++) in[k] = st.slot(kPosX + k)[j];
    }
    stamp.begin();
    // Software pipeline: the next particle's 15 loads are issued before this particle's LDS
    // atomics and density store, so an iteration never starts by waiting on memory.
    while (j < item.z) {
        const int jn = j + kTiledThreads;
        Stencil s = make_stencil(in[kPosX], in[kPosY], in[kPosZ]);
        int abase;
        const int d = ride.slot(j);
        if (ride.on) ride_carry_constants(st, ride.dst, j, d);      // stragglers included: every slot of the run
        if (stencil_in_arena(s, A, &abase))
            p2g_momentum_particle<true>(s, in[kVelX], in[kVelY], in[kVelZ], in + kC00, dens + j,
                                        dens_out + d, lds, mass_f, abase, g, P);
        else if (!LISTED)
            add_straggler(sg, sb, j);
        if (jn < item.z) {
#pragma unroll
            for (int k = 0; k < 15; k++) in[k] = st.slot(kPosX + k)[jn];
        }
        j = jn;
    }
    stamp.end(0);
    if (!LISTED) straggler_buf_flush(sg, sb);
    __syncthreads();
    flush_arena<3>(lds, A, P, g);
#if FLOWMPM_STAMPS
    __builtin_amdgcn_s_waitcnt(0);      // the flush's atomics have been accepted
    __syncthreads();
#endif
    stamp.left(0, item.z - item.y);
}

// ---- g2p (SIM:431-1536; MODES as in g2p_gather_kernel) ------------------------------

// The 27-cell gather of g2p (SIM:1322-1352): v += w*g and B += (w*g) (x) dist.  The sums are
// kept as two-float vectors -- (B0,B1), (B3,B4), (B6,B7) share the factor pair (w*g.x, w*g.y),
// (B2,B5) shares w*g.z against (dx,dy) -- so that one packed instruction does two of the
// reference's multiplications or additions; each value still sees exactly the reference's
// operations in the reference's order.
template <bool INSIDE>
__device__ __forceinline__ void g2p_gather(const Stencil &s, const float4 *grid, int abase,
                                           const float4 *__restrict__ cellf, const Params &P, Vec3 &v, float B[9])
{
    Float2v B01 = {B[0], B[1]}, B34 = {B[3], B[4]}, B67 = {B[6], B[7]}, B25 = {B[2], B[5]};
    float B8 = B[8];
    Float2v vxy = {v.x, v.y};
    float vz = v.z;
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            const float wxy = s.wx[gx] * s.wy[gy];
            const Float2v dxy = {s.dx[gx], s.dy[gy]};
            const Float2v w01 = wxy * Float2v{s.wz[0], s.wz[1]};
            const float w2 = wxy * s.wz[2];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                const float weight = gz == 0 ? w01.x : gz == 1 ? w01.y : w2;
                float4 f;
                if (INSIDE) {
                    f = grid[abase + arena_index(gx, gy, gz)];
                } else {
                    long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                    f = (c >= 0) ? cellf[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                const Float2v wvxy = Float2v{f.x, f.y} * weight;
                const float wvz = f.z * weight;
                const float dz = s.dz[gz];
                B01 = B01 + wvxy * dxy.x;
                B34 = B34 + wvxy * dxy.y;
                B67 = B67 + wvxy * dz;
                B25 = B25 + wvz * dxy;
                B8 = B8 + wvz * dz;
                vxy = vxy + wvxy;
                vz = vz + wvz;
            }
            // keep the compiler from hoisting all 27 arena reads (108 VGPRs) to the top
            if (INSIDE) asm volatile("" ::: "memory");
        }
    B[0] = B01.x; B[1] = B01.y; B[3] = B34.x; B[4] = B34.y; B[6] = B67.x; B[7] = B67.y;
    B[2] = B25.x; B[5] = B25.y; B[8] = B8;
    v.x = vxy.x; v.y = vxy.y; v.z = vz;
}

// LISTED: the momentum pass (or p2g2) of this step ran on the same particles in the same slots and
// left the list of those whose stencil is outside their arena -- g2p applies the same test to the
// same positions, so it skips them without listing them again.
template <bool MODES, bool LISTED>
__global__ void __launch_bounds__(kTiledThreads, FLOWMPM_G2P_WAVES)
g2p_tiled_kernel(TiledWork w, const float4 *__restrict__ cellf, Params P,
                 const ModeParams *__restrict__ M)
{
    const CtlHead head = load_head(w.ctl);
    const Store st{head.cur ? w.store.base[1] : w.store.base[0], w.store.capacity};
    int4 item;
    if (!take_item(w, head, &item)) return;
    const Stragglers sg = w.sg;
    ClockStamp stamp;
    stamp.entered();
    __shared__ float4 grid[kArenaCells];
    const Arena A = arena_of_tile(item.x, w.td, P);
    // The arena's cells and the first particles are requested together, the arena first: loads return
    // in order, so the arena can be written to LDS while the particle loads are still in flight, and a
    // work item waits for memory once at its start, not twice (under load a round trip takes several
    // microseconds, against about 50 for a whole work item).  0.617 -> 0.592 ms.
    constexpr int kArenaPerThread = (kArenaCells + kTiledThreads - 1) / kTiledThreads;
    float4 arena_raw[kArenaPerThread];
#pragma unroll
    for (int q = 0; q < kArenaPerThread; q++) {
        const int a = threadIdx.x + q * kTiledThreads;
        const long long c = a < kArenaCells ? arena_cell_to_grid(a, A, P) : -1;
        // always a load (of cell 0 for a cell outside the grid), then a select: a load under a branch is
        // waited for at the end of the branch
        const float4 v = cellf[c >= 0 ? c : 0];
        arena_raw[q] = (c >= 0) ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }

    // Software pipeline: the next particle's loads are issued before this particle's 18 stores
    // (the memory counter retires in order, so a load issued after them would wait for them).
    G2PIn in;
    int j = item.y + (head.riding != 0 && head.mixing != 0 ? interleaved_thread() : (int)threadIdx.x);      // as in the momentum pass
    auto load = [&](G2PIn &r, int q) {
        r.px = st.slot(kPosX)[q]; r.py = st.slot(kPosY)[q]; r.pz = st.slot(kPosZ)[q];
        // the mass is read by the mouse force alone (SIM:1354-1360)
        r.mass = P.cursor_interaction == 1u ? st.slot(kMass)[q] : 0.0f;
        r.dirx = st.slot(kDirX)[q]; r.diry = st.slot(kDirY)[q]; r.dirz = st.slot(kDirZ)[q];
    };
    int drifted = 0;    // particles this thread leaves outside the arena: stragglers of the next step
    const Ride ride = ride_of(w, head, st);
    const Store out = ride.dst;
    load(in, min(j, item.z - 1));          // unconditional, for the same reason
#pragma unroll
    for (int q = 0; q < kArenaPerThread; q++) {
        const int a = threadIdx.x + q * kTiledThreads;
        if (a < kArenaCells) grid[a] = arena_raw[q];
    }
    __syncthreads();
    stamp.begin();
    while (j < item.z) {
        const int jn = j + kTiledThreads;
        Stencil s = make_stencil(in.px, in.py, in.pz);
        int abase;
        if (stencil_in_arena(s, A, &abase)) {
            Vec3 v;
            int index = 0;
            bool hide = false;
            if (MODES) {
                index = st.ids()[j];
                hide = mode_hidden(index, M);
                v = g2p_pre_forces_modes(in.px, in.py, in.pz, st.slot(kColR)[j], st.slot(kColG)[j], st.slot(kColB)[j], index,
                                         hide, P, M);
            }
            else
                v = g2p_pre_forces(in.px, in.py, in.pz, P);
            float B[9];
#pragma unroll
            for (int k = 0; k < 9; k++) B[k] = 0.0f;
            // The gather always sits behind a branch the compiler cannot fold (every launch has tiles).  In
            // one basic block with the noise field's arithmetic the chaos kernel keeps 142 VGPRs live, three
            // waves per SIMD; in a block of its own 81, five waves: 0.620 -> 0.554 ms per launch.
            if (MODES ? (M->grid_mode == 0u && !hide) : (w.n_tiles > 0)) g2p_gather<true>(s, grid, abase, cellf, P, v, B);
            G2POut o = MODES ? g2p_finish_modes(in, v, B, index, hide, P, M) : g2p_finish(in, v, B, P);
            // While a re-bin rides on this step the record goes to its slot of the other store, and the
            // columns no kernel of the step writes travel with it (the density went there with the
            // momentum pass); the drift is then counted against the arena of the particle's new tile.
            int d = j;
            Arena An = A;
            if (ride.on) {
                const int tile = ride.key[j];
                d = ride.start[tile] + ride.rank[j];
                An = arena_of_tile(tile, w.td, P);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) out.slot(kC00 + k)[d] = o.C[k];
            out.slot(kPosX)[d] = o.px; out.slot(kPosY)[d] = o.py; out.slot(kPosZ)[d] = o.pz;
            out.slot(kVelX)[d] = o.vx; out.slot(kVelY)[d] = o.vy; out.slot(kVelZ)[d] = o.vz;
            out.slot(kDirX)[d] = o.dirx; out.slot(kDirY)[d] = o.diry; out.slot(kDirZ)[d] = o.dirz;
            const int bx = f32_to_i32(o.px) - 1 - An.ox, by = f32_to_i32(o.py) - 1 - An.oy,
                      bz = f32_to_i32(o.pz) - 1 - An.oz;
            drifted += !((unsigned)bx <= (unsigned)(kAX - 3) && (unsigned)by <= (unsigned)(kAY - 3) &&
                         (unsigned)bz <= (unsigned)(kAZ - 3));
        } else if (!LISTED) {
            add_straggler_direct(sg, j);
        }
        if (jn < item.z) load(in, jn);
        j = jn;
    }
    stamp.end(1);
#if FLOWMPM_STAMPS
    __syncthreads();
#endif
    stamp.left(1, item.z - item.y);
    if (__any(drifted != 0)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) drifted += __shfl_down(drifted, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(ride.on ? &w.ctl->drift_next : &w.ctl->cum_drift, drifted);
    }
}

// ---- stragglers --------------------------------------------------------------------
// One thread per particle the tiled kernel did not handle -- those it listed, and the live
// particles past the binned range (arrivals from other slabs since the last re-bin) -- with
// contributions straight to / from the global grids (the reference's own access pattern).
// PASS: 0 p2g1, 1 p2g2.
__device__ __forceinline__ int straggler_total(const TiledWork &w, int *count, int *tail_begin)
{
    *count = *w.sg.count;
    *tail_begin = min(w.ctl->dense, w.n);
    return *count + (w.n - *tail_begin);
}

template <int PASS>
__global__ void __launch_bounds__(256)
straggler_p2g_kernel(TiledWork w, FixedGrid g, Params P)
{
    const Store st = w.store.pick();
    int count, tail_begin;
    const int total = straggler_total(w, &count, &tail_begin);
    if (blockIdx.x == 0 && threadIdx.x == 0) *w.sg.next_count = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i < count ? w.sg.list[i] : tail_begin + (i - count);
        float in[15];
#pragma unroll
        for (int k = 0; k < 15; k++) in[k] = st.slot(kPosX + k)[j];
        Stencil s = make_stencil(in[kPosX], in[kPosY], in[kPosZ]);
        mark_dirty_stencil(s, w.td, P, g);
        if (PASS == 0) p2g1_particle<false>(s, in[kVelX], in[kVelY], in[kVelZ], in + kC00, nullptr, 0, g, P);
        if (PASS == 1) p2g2_particle<false>(s, in + kC00, st.slot(kDensity) + j, nullptr, nullptr, 0, g, P);
    }
}

__device__ __forceinline__ float pick3(const float a[3], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }

// The stragglers of the two passes of the step (MOMENTUM false: mass pass, true: momentum pass),
// 32 lanes to a particle: lane c < 27 takes stencil cell c = (gx * 3 + gy) * 3 + gz -- its mass
// word, its three momentum words -- so a particle's 27 + 81 atomics and 27 mass loads are issued
// side by side instead of one after the other, and the three cells of a z run fall into adjacent
// records.  The density is the one quantity that couples the cells: it is summed in the reference's
// order (SIM:353-364), every lane adding the 27 products one after the other.
// block, blocks: this workgroup's index among the workgroups that share the work, and their number.
template <bool MOMENTUM>
__device__ __forceinline__ void straggler_p2g_team(const TiledWork &w, const FixedGrid &g, const Params &P, int block, int blocks)
{
    const Store st = w.store.pick();
    int count, tail_begin;
    const int total = straggler_total(w, &count, &tail_begin);
    if (block == 0 && threadIdx.x == 0) *w.sg.next_count = 0;
    const int c = threadIdx.x & 31;                    // stencil cell of this lane (27..31: none)
    const int team_lane0 = threadIdx.x & 32;           // first lane of this team within the wave
    const int gx = c / 9, gy = (c / 3) % 3, gz = c % 3;
    const int teams = blocks * (blockDim.x >> 5);
    // every lane of a wave runs the same number of rounds (the shuffles below need the whole team)
    for (int i0 = block * (blockDim.x >> 5) + ((threadIdx.x >> 6) << 1); i0 < total; i0 += teams) {
        const int i = i0 + ((threadIdx.x >> 5) & 1);
        const bool live = i < total;
        const int j = !live ? 0 : (i < count ? w.sg.list[i] : tail_begin + (i - count));
        float px = 2.0f, py = 2.0f, pz = 2.0f;
        if (live) { px = st.slot(kPosX)[j]; py = st.slot(kPosY)[j]; pz = st.slot(kPosZ)[j]; }
        const Stencil s = make_stencil(px, py, pz);
        const bool mine = live && c < 27 &&
                          !(!MOMENTUM && (s.cx + 1 < w.keep_lo || s.cx + 1 >= w.keep_hi));
        const float weight = (pick3(s.wx, gx) * pick3(s.wy, gy)) * pick3(s.wz, gz);
        const long long cell = mine ? local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P) : -1;
        if (mine && c < 8 && g.dirty) {                // mark_dirty_stencil, one corner per lane
            const int cx = s.cx + (c & 4 ? 2 : 0), cy = s.cy + (c & 2 ? 2 : 0), cz = s.cz + (c & 1 ? 2 : 0);
            const int lx = cx - P.x_off;
            if ((unsigned)lx < (unsigned)P.local_nx && (unsigned)cy < (unsigned)P.gy && (unsigned)cz < (unsigned)P.gz)
                set_dirty(&g.dirty[((lx / kTX) * w.td.tiles_y + cy / kTY) * w.td.tiles_z + cz / kTZ], g.dirty_bits);
        }
        if (!MOMENTUM) {
            if (cell >= 0) atomicAdd(g.mass + cell, encode_fixed(weight));
            // a riding re-bin counts in the mass pass: the particles the tiled kernel left to this one
            if (mine && c == 0 && w.ctl->riding == 2) {         // (`mine` excludes a slab's leavers)
                const int k = tile_of(px, py, pz, P, w.td);
                w.key[j] = k;
                w.rank[j] = atomicAdd(&w.tile_count[k], 1);
            }
            continue;
        }
        float vx = 0.0f, vy = 0.0f, vz = 0.0f, C[9];
#pragma unroll
        for (int k = 0; k < 9; k++) C[k] = 0.0f;
        if (live) {
            vx = st.slot(kVelX)[j]; vy = st.slot(kVelY)[j]; vz = st.slot(kVelZ)[j];
#pragma unroll
            for (int k = 0; k < 9; k++) C[k] = st.slot(kC00 + k)[j];
        }
        const float term = decode_fixed(cell >= 0 ? g.mass[cell] : 0) * weight;
        float density = 0.0f;
        for (int k = 0; k < 27; k++) density = density + __shfl(term, team_lane0 + k);
        if (live && c == 0) {
            const Ride ride = ride_of(w, st);
            const int d = ride.slot(j);
            ride.dst.slot(kDensity)[d] = st.slot(kDensity)[j] * (1.0f - 0.05f) + density * 0.05f;      // SIM:365-366
            // a particle no work item covers: nothing else carries its constant columns to the new slot
            if (ride.on && i >= count) ride_carry_constants(st, ride.dst, j, d);
        }
        const Eq16 e = p2g2_stress(density, C, P);
        const float dx = pick3(s.dx, gx), dy = pick3(s.dy, gy), dz = pick3(s.dz, gz);
        // as in p2g_momentum_particle
        const float qx = (C[0] * dx + C[3] * dy) + C[6] * dz;
        const float qy = (C[1] * dx + C[4] * dy) + C[7] * dz;
        const float qz = (C[2] * dx + C[5] * dy) + C[8] * dz;
        int ex = encode_fixed(weight * (vx + qx));
        int ey = encode_fixed(weight * (vy + qy));
        int ez = encode_fixed(weight * (vz + qz));
        const float a00 = e.m00 * weight, a11 = e.m11 * weight, a22 = e.m22 * weight;
        const float a01 = e.m01 * weight, a02 = e.m02 * weight, a12 = e.m12 * weight;
        ex = add_wrap(ex, encode_fixed((a00 * dx + a01 * dy) + a02 * dz));
        ey = add_wrap(ey, encode_fixed((a01 * dx + a11 * dy) + a12 * dz));
        ez = add_wrap(ez, encode_fixed((a02 * dx + a12 * dy) + a22 * dz));
        if (cell >= 0) {
            atomicAdd(g.cell + 4 * cell + 0, ex);
            atomicAdd(g.cell + 4 * cell + 1, ey);
            atomicAdd(g.cell + 4 * cell + 2, ez);
        }
    }
}

template <bool MOMENTUM>
__global__ void __launch_bounds__(256)
straggler_p2g_team_kernel(TiledWork w, FixedGrid g, Params P)
{
    straggler_p2g_team<MOMENTUM>(w, g, P, blockIdx.x, gridDim.x);
    if (!MOMENTUM && w.scan_here) {                      // uniform over the launch
        // every particle of the step has been counted once the last workgroup arrives here
        if (last_workgroup(&w.ctl->ticket) && w.ctl->riding == 2) {
            const int next_set = w.ctl->item_set ^ 1;
            scan_tiles(w.tile_count, w.n_tiles, const_cast<int *>(w.tile_start) + (long long)next_set * (w.n_tiles + 1),
                       const_cast<int4 *>(w.items) + (long long)next_set * w.item_cap,
                       const_cast<int *>(w.n_items) + next_set, w.item_cap, w.chunk, w.n, w.sparse, &w.ctl->dense_next);
        }
    }
}

template <bool MODES>
__global__ void __launch_bounds__(256)
straggler_g2p_kernel(TiledWork w, const float4 *__restrict__ cellf, Params P,
                     const ModeParams *__restrict__ M)
{
    const Store st = w.store.pick();
    int count, tail_begin;
    const int total = straggler_total(w, &count, &tail_begin);
    const Ride ride = ride_of(w, st);
    const Store out = ride.dst;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *w.sg.next_count = 0;
        w.ctl->stragglers += (unsigned long long)total;
        // bookkeeping of the step that ends here: every particle on this path counts as outside
        // (not while a re-bin rides on the step: these particles are being binned anew)
        // (the particles of nearly empty tiles are on this path by choice: they do not count)
        if (!ride.on) {
            const int outside = count + (w.n - min(w.ctl->binned, w.n));
            if (outside) atomicAdd(&w.ctl->cum_drift, outside);
            w.ctl->age += 1;
        }
    }
    int drifted = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i < count ? w.sg.list[i] : tail_begin + (i - count);
        G2PIn in;
        in.px = st.slot(kPosX)[j]; in.py = st.slot(kPosY)[j]; in.pz = st.slot(kPosZ)[j];
        in.mass = P.cursor_interaction == 1u ? st.slot(kMass)[j] : 0.0f;
        in.dirx = st.slot(kDirX)[j]; in.diry = st.slot(kDirY)[j]; in.dirz = st.slot(kDirZ)[j];
        Vec3 v;
        int index = 0;
        bool hide = false;
        if (MODES) {
            index = st.ids()[j];
            hide = mode_hidden(index, M);
            v = g2p_pre_forces_modes(in.px, in.py, in.pz, st.slot(kColR)[j], st.slot(kColG)[j], st.slot(kColB)[j], index,
                                     hide, P, M);
        }
        else
            v = g2p_pre_forces(in.px, in.py, in.pz, P);
        Stencil s = make_stencil(in.px, in.py, in.pz);
        float B[9];
#pragma unroll
        for (int k = 0; k < 9; k++) B[k] = 0.0f;
        if (!MODES || (M->grid_mode == 0u && !hide)) g2p_gather<false>(s, nullptr, 0, cellf, P, v, B);
        G2POut o = MODES ? g2p_finish_modes(in, v, B, index, hide, P, M) : g2p_finish(in, v, B, P);
        int d = j;
        if (ride.on) {            // as in g2p_tiled_kernel
            const int tile = ride.key[j];
            d = ride.start[tile] + ride.rank[j];
            const Arena An = arena_of_tile(tile, w.td, P);
            const int bx = f32_to_i32(o.px) - 1 - An.ox, by = f32_to_i32(o.py) - 1 - An.oy,
                      bz = f32_to_i32(o.pz) - 1 - An.oz;
            drifted += !((unsigned)bx <= (unsigned)(kAX - 3) && (unsigned)by <= (unsigned)(kAY - 3) &&
                         (unsigned)bz <= (unsigned)(kAZ - 3));
        }
#pragma unroll
        for (int k = 0; k < 9; k++) out.slot(kC00 + k)[d] = o.C[k];
        out.slot(kPosX)[d] = o.px; out.slot(kPosY)[d] = o.py; out.slot(kPosZ)[d] = o.pz;
        out.slot(kVelX)[d] = o.vx; out.slot(kVelY)[d] = o.vy; out.slot(kVelZ)[d] = o.vz;
        out.slot(kDirX)[d] = o.dirx; out.slot(kDirY)[d] = o.diry; out.slot(kDirZ)[d] = o.dirz;
    }
    if (!ride.on) return;                                   // uniform over the launch
    if (drifted) atomicAdd(&w.ctl->drift_next, drifted);
    // The riding re-bin is complete when every workgroup of this launch -- the last one of the step --
    // has stored its records: the last one to finish makes the other store and item set current.
    if (last_workgroup(&w.ctl->ticket) && threadIdx.x == 0) {
        RebinCtl *ctl = w.ctl;
        ctl->cur ^= 1;
        ctl->item_set ^= 1;
        ctl->age = 1;               // the new binning is of the positions before this step's g2p
        ctl->cum_drift = __hip_atomic_load(&ctl->drift_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl->drift_next = 0;
        ctl->binned = w.n;
        ctl->dense = ctl->dense_next;
        ctl->rebins += 1;
        ctl->epoch += 1;
        ctl->riding = 0;
    }
}

// ---- clearGrid and updateGrid over the dirty tiles only --------------------------------
//
// FixedGrid::dirty says which tiles may hold a non-zero word in which buffer.  One workgroup per
// tile; a tile whose bits are clear costs one load.  Tiles with tx <= always_lo or
// tx >= always_hi are treated as dirty whatever their bits: a slab's boundary planes also
// receive the neighbours' contributions, which set no bits.

struct TileRange { int always_lo, always_hi; };

__device__ __forceinline__ bool tile_always(int tile, TileDims td, TileRange r)
{
    const int tx = tile / (td.tiles_z * td.tiles_y);
    return tx <= r.always_lo || tx >= r.always_hi;
}

// Local linear index of cell k (0..511) of a tile, or -1 outside the grid.
__device__ __forceinline__ long long tile_cell(int tile, int k, TileDims td, const Params &P, int *lx, int *y, int *z)
{
    const int tz = tile % td.tiles_z, ty = (tile / td.tiles_z) % td.tiles_y, tx = tile / (td.tiles_z * td.tiles_y);
    *z = tz * kTZ + k % kTZ;
    *y = ty * kTY + (k / kTZ) % kTY;
    *lx = tx * kTX + k / (kTZ * kTY);
    if (*lx >= P.local_nx || *y >= P.gy || *z >= P.gz) return -1;
    return ((long long)*lx * P.gy + *y) * P.gz + *z;
}

// clearGrid of the fixed-point grid (SIM:262-277 without the float grid): the momentum records `cell`
// of tiles with `cell_bit` set, the words of mass buffer `mass` of tiles with `mass_bit` set; both
// bits are cleared.
__global__ void __launch_bounds__(256)
clear_tiles_kernel(int4 *__restrict__ cell, int *__restrict__ mass, int *__restrict__ dirty, int cell_bit, int mass_bit,
                   TileDims td, Params P, TileRange r)
{
    const int t = blockIdx.x;
    const bool always = tile_always(t, td, r);
    const int bits = dirty[t];
    const bool do_cell = always || (bits & cell_bit), do_mass = always || (bits & mass_bit);
    if (!do_cell && !do_mass) return;
    for (int k = threadIdx.x; k < kTX * kTY * kTZ; k += 256) {
        int lx, y, z;
        const long long c = tile_cell(t, k, td, P, &lx, &y, &z);
        if (c < 0) continue;
        if (do_cell) cell[c] = make_int4(0, 0, 0, 0);
        if (do_mass) mass[c] = 0;
    }
    __syncthreads();                       // every thread has read the bits
    if (threadIdx.x == 0) dirty[t] = bits & ~(cell_bit | mass_bit);
}

// updateGrid (SIM:395-429, see update_grid_kernel) over the tiles where this step's mass buffer
// or the one that describes the float grid may be non-zero (`bits`).
//
// The fixed-point grid is double-buffered: a step accumulates into one set {cell, mass_i} while the set
// of the step before, {cell_old, mass_of_cellf}, still tells this kernel which float cells the fluid has
// left.  Once this kernel has read it that older set is dead, so -- clear_old_bits != 0 -- it is zeroed
// here, in the tiles whose bits say it may hold something (the same tiles this launch visits anyway), and
// the NEXT step accumulates into it without a clearGrid launch of its own.
// A workgroup takes RUN tiles that follow each other along z (consecutive tile ids).  With RUN = 4, per
// (x, y) row of the run its lanes cover up to 32 consecutive cells -- 512 contiguous bytes of momentum
// records instead of the 128 of one tile -- and every thread has the words of its 8 cells in flight at once
// (a tile per workgroup with a cell's loads one after the other moved 3 TB/s at 256^3; 0.100 -> 0.086 ms).
// A tile of the run that needs no visit idles its share of the lanes.  RUN = 1 (a workgroup per tile, two
// cells per thread) is for grids of few tiles, where more workgroups matter more than longer rows.
static_assert(kTX == 8 && kTY == 8 && kTZ == 8, "the lane mapping below is for 8x8x8 tiles");
constexpr int kTileRunMax = 4;
constexpr int kTileRunFromTiles = 4096;      // grids of more tiles than this take runs of kTileRunMax
inline int tile_run(int n_tiles) { return n_tiles > kTileRunFromTiles ? kTileRunMax : 1; }
inline int tile_runs_z(int tiles_z, int run) { return (tiles_z + run - 1) / run; }

template <int RUN>
__global__ void __launch_bounds__(256)
update_grid_tiles_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass_i,
                         int *__restrict__ mass_of_cellf, float4 *__restrict__ cellf,
                         int *__restrict__ dirty, int bits, TileDims td, Params P, TileRange r,
                         int4 *__restrict__ cell_old, int clear_old_bits)
{
    static_assert(RUN == 1 || RUN == 2 || RUN == 4, "256 threads = 8 z x RUN tiles x 8 y x (4 / RUN) groups of planes");
    constexpr int kRunBits = RUN == 4 ? 2 : (RUN == 2 ? 1 : 0);
    constexpr int kPer = kTX * RUN / 4;             // planes (cells) per thread
    // block -> (tile column tx, ty, run of z tiles)
    const int runs_z = (td.tiles_z + RUN - 1) / RUN;
    const int per_column = td.tiles_y * runs_z;
    const int tx = blockIdx.x / per_column;
    const int rest = blockIdx.x - tx * per_column;
    const int ty = rest / runs_z, run = rest - ty * runs_z;
    const int zc = threadIdx.x & 7, q = (threadIdx.x >> 3) & (RUN - 1), row = (threadIdx.x >> (3 + kRunBits)) & 7,
              group = threadIdx.x >> (6 + kRunBits);
    const int tz = run * RUN + q;
    const bool exists = tz < td.tiles_z;
    const int t = (tx * td.tiles_y + ty) * td.tiles_z + tz;
    const bool always = tx <= r.always_lo || tx >= r.always_hi;
    const int tile_bits = exists ? dirty[t] : 0;
    const bool zero_old = exists && clear_old_bits != 0 && (always || (tile_bits & clear_old_bits));
    const bool visit = exists && (always || zero_old || (tile_bits & bits));
    const int y = ty * kTY + row, z = tz * kTZ + zc, lx0 = tx * kTX + group * kPer;
    if (visit && y < P.gy && z < P.gz && lx0 < P.local_nx) {
        const long long plane = (long long)P.gy * P.gz;
        const long long first = ((long long)lx0 * P.gy + y) * P.gz + z;
        const int nx = min(kPer, P.local_nx - lx0);               // this thread's planes inside the grid
        int4 c[kPer];
        int m[kPer], before[kPer];
#pragma unroll
        for (int p = 0; p < kPer; p++) {
            const long long i = first + (p < nx ? p : 0) * plane;     // (a load outside the grid repeats the first cell's)
            m[p] = mass_i[i];
            before[p] = mass_of_cellf[i];
            c[p] = cell[i];
        }
#pragma unroll
        for (int p = 0; p < kPer; p++) {
            if (p >= nx) break;
            const long long i = first + p * plane;
            update_grid_cell_loaded(c[p], m[p], before[p], cellf, i, lx0 + p + P.x_off, y, z, P);
            if (zero_old) {
                cell_old[i] = make_int4(0, 0, 0, 0);
                mass_of_cellf[i] = 0;
            }
        }
    }
    __syncthreads();                       // every thread has read the bits
    if (zero_old && zc == 0 && row == 0 && group == 0) dirty[t] = tile_bits & ~clear_old_bits;
}

// ---- host side -------------------------------------------------------------------

inline TileDims tile_dims(const TiledState &T) { return TileDims{T.tiles_x, T.tiles_y, T.tiles_z}; }

// The tile dimensions of a local grid of local_nx planes (a slab that is re-cut calls this again).
inline void tiled_set_planes(TiledState &T, const Params &P, int local_nx)
{
    T.tiles_x = (local_nx + kTX - 1) / kTX;
    T.tiles_y = (P.gy + kTY - 1) / kTY;
    T.tiles_z = (P.gz + kTZ - 1) / kTZ;
    T.n_tiles = T.tiles_x * T.tiles_y * T.tiles_z;
}

// max_local_nx: the most planes the local grid will ever hold (the per-tile arrays are sized for it).
inline int tiled_init(TiledState &T, const Params &P, long long capacity, int max_local_nx, hipStream_t stream,
                      std::string *msg)
{
    tiled_set_planes(T, P, max_local_nx);
    T.tiles_cap = T.n_tiles;
    T.item_cap = (long long)T.n_tiles + capacity / kChunkMin + 1;
    hipError_t e;
    auto chk = [&](hipError_t err, const char *what) {
        if (err != hipSuccess) { *msg = std::string(what) + ": " + hipGetErrorString(err); return true; }
        return false;
    };
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.key), (size_t)capacity * sizeof(int)), "hipMalloc key")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.rank), (size_t)capacity * sizeof(int)), "hipMalloc rank")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.tile_count), (size_t)T.n_tiles * sizeof(int)), "hipMalloc tile_count")) return 1;
    if (chk(e = hipMemsetAsync(T.tile_count, 0, (size_t)T.n_tiles * sizeof(int), stream), "memset tile_count")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.tile_start), 2 * (size_t)(T.n_tiles + 1) * sizeof(int)), "hipMalloc tile_start")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.items), 2 * (size_t)T.item_cap * sizeof(int4)), "hipMalloc items")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.straggler_counts), 256), "hipMalloc straggler counts")) return 1;
    if (chk(e = hipMemsetAsync(T.straggler_counts, 0, 256, stream), "memset straggler counts")) return 1;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.straggler_list), (size_t)capacity * sizeof(int)), "hipMalloc straggler list")) return 1;
    T.stage_parity = 0;
    if (chk(e = hipMalloc(reinterpret_cast<void **>(&T.ctl), sizeof(RebinCtl)), "hipMalloc rebin control")) return 1;
    if (chk(e = hipMemsetAsync(T.ctl, 0, sizeof(RebinCtl), stream), "memset rebin control")) return 1;
    if (chk(e = hipMemsetAsync(&T.ctl->t_start, 0xFF, sizeof(unsigned long long), stream), "memset rebin clock")) return 1;
    T.n_items = reinterpret_cast<int *>(reinterpret_cast<char *>(T.ctl) + offsetof(RebinCtl, n_items));
    T.valid = false;
    T.checked = false;
    tiled_set_planes(T, P, P.local_nx);
    return 0;
}

inline void tiled_free(TiledState &T)
{
    if (T.key) (void)hipFree(T.key);
    if (T.rank) (void)hipFree(T.rank);
    if (T.tile_count) (void)hipFree(T.tile_count);
    if (T.tile_start) (void)hipFree(T.tile_start);
    if (T.items) (void)hipFree(T.items);
    if (T.straggler_counts) (void)hipFree(T.straggler_counts);
    if (T.straggler_list) (void)hipFree(T.straggler_list);
    if (T.ctl) (void)hipFree(T.ctl);
    T = TiledState{};
}

// Blocks of the grid-stride binning kernels: enough to fill the chip, few enough that a
// launch which finds no re-bin due costs next to nothing.
#ifndef FLOWMPM_BIN_BLOCKS
#define FLOWMPM_BIN_BLOCKS 2048
#endif
constexpr unsigned kBinBlocks = FLOWMPM_BIN_BLOCKS;

// Work-item size of a binning.  Measured (profiles/r02_chunk_sweep.txt): splitting a tile over more,
// smaller work items to fill the chip makes small scenes SLOWER -- every item pays for its arena (zeroing,
// the mass words, the flush), and that, not idle compute units, is what a small scene spends its time on
// (2^20 particles / 128^3: 4096 -> 5.51e9/s, 512 -> 4.94, 256 -> 3.85).  So the upper bound is the
// default; TiledState::chunk_wanted (flowmpm_set_option) asks for another size, for experiments.
inline int choose_chunk(const TiledState &T)
{
    const int chunk = (T.chunk_wanted + kTiledThreads - 1) / kTiledThreads * kTiledThreads;
    return std::min(kChunkMax, std::max(kChunkMin, chunk));
}

// Below how many particles a tile gets no work item (RebinCtl::dense); 0 gives every occupied tile one.
inline int choose_sparse(const TiledState &T) { return std::min(std::max(T.sparse_wanted, 0), kChunkMin); }

// Enqueues the start-of-step pair of launches (see "binning" above).  When a re-bin is due the
// first n particles of the current store are grouped by tile into the other store, the work
// items are rebuilt and the other store becomes current.
//
// ride: only the first launch is enqueued; a re-bin it finds due is completed by the momentum pass and
// the g2p stage of the same step, which must follow (flowmpm_step's order): they store what they store
// anyway into the other store, at the particle's new slot, so the records are not moved a second time.
// A forced re-bin (new particles: no binning to run the step's first passes on) never rides.
// ride: 0 no; 1 the counting launch of its own; 2 the mass pass of the step counts (it must be the next
// tiled stage: launch_p2g_mass_tiled then enqueues the launch that builds the next item set).
inline int rebin_if_due(TiledState &T, StoreSel ss, long long n, const Params &P, int interval, int divisor,
                        hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, std::string *msg,
                        int ride = 0)
{
    const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, kBinBlocks);
    const int force = T.valid ? 0 : 1;
    if (force) {
        ride = 0;
        T.chunk = choose_chunk(T);      // kept until the next forced re-bin: the launch bounds rely on it
        T.sparse = choose_sparse(T);
    }
    T.listed = -1;
    if (ride == 2) {
        T.ride_interval = interval;
        T.ride_divisor = divisor;
        T.scan_after_mass = true;
        T.scan_ev_start = ev_start;
        T.scan_ev_stop = ev_stop;
        T.checked = true;
        return 0;
    }
    hipExtLaunchKernelGGL(bin_count_scan_kernel, dim3(blocks), dim3(256), 0, stream, ev_start, ride ? ev_stop : nullptr,
                          0, ss, (int)n, P, tile_dims(T), T.key, T.rank, T.tile_count, T.n_tiles, T.tile_start, T.items,
                          T.n_items, T.item_cap, T.ctl, interval, force, divisor, ride ? 1 : 0, T.chunk, T.sparse);
    if (!ride)
        hipExtLaunchKernelGGL(bin_scatter_kernel, dim3(blocks), dim3(256), 0, stream, nullptr, ev_stop, 0, ss, (int)n,
                              T.key, T.rank, T.tile_start, T.n_tiles, T.ctl, interval, force, divisor);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *msg = std::string("rebin launch: ") + hipGetErrorString(e); return 1; }
    T.valid = true;
    T.checked = true;
    return 0;
}

// The work description of the next stage.  Stages alternate between two straggler counters:
// the follow-up kernel of a stage zeroes the one the next stage will count into, which saves a
// memset launch per stage.
// A stage launched in two parts (TiledWork::part) counts its stragglers in one counter: part 1
// leaves the choice of counter to part 2, whose follow-up kernel handles what both listed.
struct StagePart { int part, bnd_lo, bnd_hi; };
constexpr StagePart kWholeStage{0, 0, 0};

inline TiledWork tiled_work(TiledState &T, StoreSel ss, long long n, StagePart sp = kWholeStage)
{
    const Stragglers sg{T.straggler_counts + T.stage_parity, T.straggler_counts + (T.stage_parity ^ 1),
                        T.straggler_list};
    if (sp.part != 1) T.stage_parity ^= 1;
    return TiledWork{ss, T.items, T.n_items, T.item_cap, T.key, T.rank, T.tile_count, T.tile_start, T.n_tiles, T.ctl, (int)n,
                     tile_dims(T), sg, -2147483647 - 1, 2147483647, sp.part, sp.bnd_lo, sp.bnd_hi, 0, 0, 0, 0, 0, 0, 0};
}

// Upper bound of the work items any binning of n particles can produce.
inline unsigned item_bound(const TiledState &T, long long n)
{
    long long b = std::min<long long>(T.item_cap, std::min<long long>((long long)T.n_tiles, n) + n / T.chunk + 1);
    return (unsigned)std::max<long long>(b, 1);
}

// Every tiled stage is: the tiled kernel, then the straggler kernel for what it left.
// ev_start / ev_stop (may be null) receive the start and end timestamps of the tiled kernel.
// Workgroups of a follow-up kernel: a small scene's handful of stragglers does not need a thousand.
constexpr int kStragglerBlocksMax = 1024;
inline unsigned straggler_blocks(long long n)
{
    return (unsigned)std::min<long long>(kStragglerBlocksMax, std::max<long long>(32, n / 4096));
}

#define FLOWMPM_TILED_STAGE(kernel, straggler_kernel, grid_ptr)                                              \
    do {                                                                                                    \
        const TiledWork w = tiled_work(T, st, n);                                                           \
        hipExtLaunchKernelGGL(kernel, dim3(item_bound(T, n)), dim3(kTiledThreads), 0, stream, ev_start,     \
                              ev_stop, 0, w, grid_ptr, P);                                                  \
        hipLaunchKernelGGL(straggler_kernel, dim3(straggler_blocks(n)), dim3(256), 0, stream, w, grid_ptr, P); \
        return 0;                                                                                           \
    } while (0)

inline int launch_p2g1_tiled(TiledState &T, StoreSel st, long long n, FixedGrid g, const Params &P, hipStream_t stream,
                             hipEvent_t ev_start, hipEvent_t ev_stop)
{
    T.listed = -1;      // this stage overwrites the one straggler list: a momentum pass's list is gone
    FLOWMPM_TILED_STAGE(p2g1_tiled_kernel, straggler_p2g_kernel<0>, g);
}

inline int launch_p2g2_tiled(TiledState &T, StoreSel st, long long n, FixedGrid g, const Params &P, hipStream_t stream,
                             hipEvent_t ev_start, hipEvent_t ev_stop)
{
    T.listed = T.stage_parity;
    FLOWMPM_TILED_STAGE(p2g2_tiled_kernel, straggler_p2g_kernel<1>, g);
}

// keep_lo / keep_hi: see TiledWork.
// The launch that turns the tile counts of a riding re-bin into the next item set (a no-op on the device
// when the mass pass found no re-bin due).  n: the particles counted.
inline void launch_ride_scan(TiledState &T, long long n, hipStream_t stream)
{
    if (!T.scan_after_mass) return;
    T.scan_after_mass = false;
    hipExtLaunchKernelGGL(ride_scan_kernel, dim3(1), dim3(1024), 0, stream, T.scan_ev_start, T.scan_ev_stop, 0, T.ctl,
                          T.tile_count, T.n_tiles, T.tile_start, T.items, T.n_items, T.item_cap, T.chunk, (int)n,
                          T.sparse);
}

// defer_scan: the caller launches the scan itself (launch_ride_scan), after it has counted the particles
// this pass does not see (a slab's arrivals).  Otherwise a grid of at most kScanInFollowUpTiles tiles is
// scanned by the last workgroup of the pass's follow-up kernel (256 threads: fine for a few thousand tiles,
// too slow for the 32,768 of a 256^3 grid, which keep the 1,024-thread launch).
constexpr int kScanInFollowUpTiles = 4096;
inline int launch_p2g_mass_tiled(TiledState &T, StoreSel st, long long n, FixedGrid g, const Params &P, hipStream_t stream,
                                 hipEvent_t ev_start, hipEvent_t ev_stop, int keep_lo = -2147483647 - 1,
                                 int keep_hi = 2147483647, bool defer_scan = false)
{
    // This stage overwrites the one straggler list.  Its own list serves the momentum pass and g2p of the step
    // as long as the particles stay where they are -- not when it skips a slab's leavers, whose slots other
    // particles take before the momentum pass runs.
    const bool everything = keep_lo == -2147483647 - 1 && keep_hi == 2147483647;
    T.listed = everything ? T.stage_parity : -1;
    TiledWork w = tiled_work(T, st, n);
    w.keep_lo = keep_lo;
    w.keep_hi = keep_hi;
    if (T.scan_after_mass) {
        w.decide = 1;
        w.interval = T.ride_interval;
        w.divisor = T.ride_divisor;
        if (!defer_scan && T.n_tiles <= kScanInFollowUpTiles) {
            w.scan_here = 1;
            w.chunk = T.chunk;
            w.sparse = T.sparse;
            T.scan_after_mass = false;           // no launch of its own
        }
    }
    hipExtLaunchKernelGGL(p2g_mass_tiled_kernel, dim3(item_bound(T, n)), dim3(kTiledThreads), 0, stream, ev_start,
                          ev_stop, 0, w, g, P);
    // (with the scan inside it, the follow-up kernel is what the re-bin's timing events bracket)
    hipExtLaunchKernelGGL(straggler_p2g_team_kernel<false>, dim3(straggler_blocks(n)), dim3(256), 0, stream,
                          w.scan_here ? T.scan_ev_start : nullptr, w.scan_here ? T.scan_ev_stop : nullptr, 0, w, g, P);
    if (!defer_scan) launch_ride_scan(T, n, stream);
    return 0;
}

// The mass pass for the particles in slots [first, n) alone (a slab's arrivals, inserted after
// its mass pass ran), straight to the global grid.
// While a re-bin rides on the step whose mass pass counts the tiles (RebinCtl::riding == 2) these
// particles are counted here: they arrived after that pass.
__global__ void __launch_bounds__(256)
p2g_mass_range_kernel(StoreSel ss, int first, int n, FixedGrid g, Params P, TileDims td, const RebinCtl *ctl,
                      int *__restrict__ key, int *__restrict__ rank, int *__restrict__ tile_count)
{
    const Store st = ss.pick();
    const bool counting = ctl->riding == 2;
    for (int j = first + blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float px = st.slot(kPosX)[j], py = st.slot(kPosY)[j], pz = st.slot(kPosZ)[j];
        Stencil s = make_stencil(px, py, pz);
        mark_dirty_stencil(s, td, P, g);
        p2g_mass_particle<false>(s, nullptr, 0, g, P);
        if (counting) {
            const int k = tile_of(px, py, pz, P, td);
            key[j] = k;
            rank[j] = atomicAdd(&tile_count[k], 1);
        }
    }
}

inline int launch_p2g_momentum_tiled(TiledState &T, StoreSel st, long long n, FixedGrid g, const Params &P,
                                     hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop,
                                     StagePart sp = kWholeStage)
{
    if (T.listed >= 0 && sp.part == 0) {
        // the mass pass's list: one launch, the follow-up work in its first workgroups; the stage counts into
        // no counter of its own (as in launch_g2p_tiled)
        TiledWork w = tiled_work(T, st, n, StagePart{1, 0, 0});      // part 1: no change of counter
        w.part = 0;
        w.sg = Stragglers{T.straggler_counts + T.listed, T.straggler_counts + (T.listed ^ 1), T.straggler_list};
        w.follow_blocks = (int)straggler_blocks(n);
        hipExtLaunchKernelGGL(p2g_momentum_tiled_kernel<true>, dim3(item_bound(T, n) + w.follow_blocks), dim3(kTiledThreads),
                              0, stream, ev_start, ev_stop, 0, w, g, P);
        return 0;
    }
    if (sp.part != 1) T.listed = T.stage_parity;      // with part 2 the list is complete
    const TiledWork w = tiled_work(T, st, n, sp);
    hipExtLaunchKernelGGL(p2g_momentum_tiled_kernel<false>, dim3(item_bound(T, n)), dim3(kTiledThreads), 0, stream,
                          ev_start, ev_stop, 0, w, g, P);
    if (sp.part != 1)
        hipLaunchKernelGGL(straggler_p2g_team_kernel<true>, dim3(straggler_blocks(n)), dim3(256), 0, stream, w, g, P);
    return 0;
}

// modes == nullptr: chaos path.
inline int launch_g2p_tiled(TiledState &T, StoreSel st, long long n, const float4 *cellf, const Params &P,
                            const ModeParams *modes, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop,
                            StagePart sp = kWholeStage)
{
    const dim3 grid(item_bound(T, n)), block(kTiledThreads);
    // With a list of this step's stragglers at hand (TiledState::listed) the stage counts into no
    // counter of its own: its follow-up kernel reads that list, and the next stage finds the counters
    // as the momentum pass left them.
    const bool listed = T.listed >= 0;
    TiledWork w;
    if (listed) {
        w = tiled_work(T, st, n, StagePart{1, sp.bnd_lo, sp.bnd_hi});      // part 1: no change of counter
        w.part = sp.part;
        w.sg = Stragglers{T.straggler_counts + T.listed, T.straggler_counts + (T.listed ^ 1), T.straggler_list};
    } else {
        w = tiled_work(T, st, n, sp);
    }
#define FLOWMPM_G2P_LAUNCH(MODES_, LISTED_)                                                                       \
    do {                                                                                                          \
        hipExtLaunchKernelGGL((g2p_tiled_kernel<MODES_, LISTED_>), grid, block, 0, stream, ev_start, ev_stop, 0,  \
                              w, cellf, P, modes);                                                                \
        if (sp.part != 1)                                                                                         \
            hipLaunchKernelGGL(straggler_g2p_kernel<MODES_>, dim3(straggler_blocks(n)), dim3(256), 0, stream, w,  \
                               cellf, P, modes);                                                                  \
    } while (0)
    if (modes) {
        if (listed) FLOWMPM_G2P_LAUNCH(true, true); else FLOWMPM_G2P_LAUNCH(true, false);
    } else {
        if (listed) FLOWMPM_G2P_LAUNCH(false, true); else FLOWMPM_G2P_LAUNCH(false, false);
    }
#undef FLOWMPM_G2P_LAUNCH
    if (sp.part != 1) T.listed = -1;                  // the particles have moved
    return 0;
}

}  // namespace mpm
