// Particle migration between x-slabs (no reference counterpart: the reference
// runs on one GPU).  After g2p a particle whose floor(position.x) left the slab's
// owned planes is packed into a 24-word record, grouped by destination rank,
// and removed from the store; the holes are filled from the tail so the
// resident particles stay dense in slots [0, resident).
#pragma once

#include <string>
#include <vector>

#include "kernels_scatter.h"

namespace mpm {

constexpr int kMaxRanks = 64;
constexpr int kRecordWords = 24;   // slots kPosX..kColB, then the id

struct RankTable {
    int starts[kMaxRanks + 1];     // rank r owns planes [starts[r], starts[r+1])
    int n;
};
struct RankOffsets { long long off[kMaxRanks]; };

__device__ __forceinline__ bool is_leaver(float px, int x0, int nx)
{
    int ix = f32_to_i32(px);
    return ix < x0 || ix >= x0 + nx;
}

// counters[0..kMaxRanks): leavers per destination; [kMaxRanks]: total; [kMaxRanks + 1], [+ 2]: holes and
// movers of the compaction; [kRedoWord]: whether the classification in flight runs; [kEpochWord]: the
// store's binning epoch the leaver list belongs to.
constexpr int kRedoWord = kMaxRanks + 3, kEpochWord = kMaxRanks + 4, kCounterWords = kMaxRanks + 5;

// One workgroup, before find_leavers_kernel.  only_if_moved: the list of the last classification is
// kept (and the classification skipped) when the store has not been re-binned since -- `epoch` counts
// the re-bins and is null for a store that never is.
__global__ void __launch_bounds__(128)
find_prologue_kernel(int *counters, const int *epoch, int only_if_moved)
{
    const int now = epoch ? *epoch : 0;
    const bool redo = !only_if_moved || !epoch || counters[kEpochWord] != now;
    __syncthreads();                                   // every thread has read the old epoch word
    if (redo)
        for (int i = threadIdx.x; i < kMaxRanks + 3; i += blockDim.x) counters[i] = 0;
    if (threadIdx.x == 0) { counters[kRedoWord] = redo ? 1 : 0; counters[kEpochWord] = now; }
}

// Every leaver needs a place in the list and a rank among those bound for its destination: three
// counters for the whole launch.  Atomics on one word are served one after the other and each waits
// for its return value, so a workgroup collects its leavers in LDS first -- it reads one contiguous
// run of the store -- and reserves its share of each counter with one atomic.  What does not fit the
// buffer takes the counters directly.
constexpr int kFindBuf = 1024;

__global__ void __launch_bounds__(256)
find_leavers_kernel(StoreSel ss, long long n, int x0, int nx, RankTable rt, int *counters,
                    int *list_slot, int *list_dest, int *list_k, long long list_cap)
{
    if (counters[kRedoWord] == 0) return;
    __shared__ int buf_slot[kFindBuf], buf_dest_k[kFindBuf];     // destination | rank within this workgroup << 6
    __shared__ int n_buf, per_dest[kMaxRanks], base_dest[kMaxRanks], base_e;
    static_assert(kMaxRanks <= 64, "the destination is packed into 6 bits");
    if (threadIdx.x == 0) n_buf = 0;
    if (threadIdx.x < kMaxRanks) per_dest[threadIdx.x] = 0;
    __syncthreads();
    const Store st = ss.pick();
    const long long per = (((n + gridDim.x - 1) / gridDim.x) + 255) / 256 * 256;
    const long long begin = (long long)blockIdx.x * per, end = n < begin + per ? n : begin + per;
    // four loads in flight per thread; a slot past the run reads as a position inside the slab
    const float inside = (float)x0 + 0.5f;
    for (long long j0 = begin + threadIdx.x; j0 < end; j0 += 4 * blockDim.x) {
        float x4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long q = j0 + (long long)u * blockDim.x;
            x4[u] = q < end ? st.slot(kPosX)[q] : inside;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long j = j0 + (long long)u * blockDim.x;
            const float px = x4[u];
            if (!is_leaver(px, x0, nx)) continue;
            const int ix = f32_to_i32(px);
            int dest = 0;
            for (int r = 0; r < rt.n; r++)
                if (ix >= rt.starts[r]) dest = r;
            const int i = atomicAdd(&n_buf, 1);
            if (i < kFindBuf) {
                buf_slot[i] = (int)j;
                buf_dest_k[i] = dest | (atomicAdd(&per_dest[dest], 1) << 6);
            } else {
                const int k = atomicAdd(&counters[dest], 1);
                const int e = atomicAdd(&counters[kMaxRanks], 1);
                if (e < list_cap) { list_slot[e] = (int)j; list_dest[e] = dest; list_k[e] = k; }
            }
        }
    }
    __syncthreads();
    const int cnt = n_buf < kFindBuf ? n_buf : kFindBuf;
    if (cnt == 0) return;                                  // uniform
    if (threadIdx.x < rt.n && per_dest[threadIdx.x] > 0)
        base_dest[threadIdx.x] = atomicAdd(&counters[threadIdx.x], per_dest[threadIdx.x]);
    if (threadIdx.x == 128) base_e = atomicAdd(&counters[kMaxRanks], cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int e = base_e + i, dest = buf_dest_k[i] & 63;
        if (e < list_cap) { list_slot[e] = buf_slot[i]; list_dest[e] = dest; list_k[e] = base_dest[dest] + (buf_dest_k[i] >> 6); }
    }
}

__global__ void __launch_bounds__(256)
pack_leavers_kernel(StoreSel ss, const int *list_slot, const int *list_dest, const int *list_k,
                    int total, RankOffsets ro, float *__restrict__ out)
{
    const Store st = ss.pick();
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long long j = list_slot[e];
    float *r = out + (ro.off[list_dest[e]] + list_k[e]) * kRecordWords;
#pragma unroll
    for (int w = 0; w < kRecordWords; w++) r[w] = st.slot(w)[j];
}

// Holes: leaver slots below the new resident count.  Movers: stayers at or above it.
__global__ void __launch_bounds__(256)
find_holes_kernel(const int *list_slot, int total, long long new_n, int *counters, int *holes)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int j = list_slot[e];
    if (j < new_n) holes[atomicAdd(&counters[kMaxRanks + 1], 1)] = j;
}

__global__ void __launch_bounds__(256)
find_movers_kernel(StoreSel ss, long long new_n, long long n, int x0, int nx, int *counters, int *movers)
{
    const Store st = ss.pick();
    long long j = new_n + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (!is_leaver(st.slot(kPosX)[j], x0, nx)) movers[atomicAdd(&counters[kMaxRanks + 2], 1)] = (int)j;
}

// riding (may be null): a device word that is 2 while a re-bin rides on the step and its mass pass has
// already counted these particles; their tile key and rank (per slot) then move with them.
__global__ void __launch_bounds__(256)
fill_holes_kernel(StoreSel ss, const int *holes, const int *movers, const int *counters, const int *riding,
                  int *__restrict__ key, int *__restrict__ rank)
{
    const Store st = ss.pick();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n_holes = counters[kMaxRanks + 1];
    if (i >= n_holes) return;
    long long dst = holes[i], src = movers[i];
#pragma unroll
    for (int w = 0; w < kRecordWords; w++) st.slot(w)[dst] = st.slot(w)[src];
    if (riding && *riding == 2) { key[dst] = key[src]; rank[dst] = rank[src]; }
}

__global__ void __launch_bounds__(256)
unpack_records_kernel(const float *__restrict__ in, StoreSel ss, long long first_slot, long long count)
{
    const Store st = ss.pick();
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float *r = in + i * kRecordWords;
#pragma unroll
    for (int w = 0; w < kRecordWords; w++) st.slot(w)[first_slot + i] = r[w];
}

// counters[0 .. n_ranks) as 64-bit words in a caller's device buffer (what the count exchange sends).
__global__ void export_counts_kernel(const int *__restrict__ counters, int n_ranks, long long *__restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_ranks) out[r] = counters[r];
}

// Slab boundaries that give every rank about the same number of particles, from the particles-per-plane
// histogram of the whole domain: cut at the particle quantiles, every slab at least 2 planes wide
// (flow_amd/slab.py: plane_starts_from_histogram is the same rule).
inline std::vector<int> balanced_plane_starts(const std::vector<int64_t> &hist, int world)
{
    const int gx = (int)hist.size();
    std::vector<int64_t> cum((size_t)gx);
    int64_t total = 0;
    for (int x = 0; x < gx; x++) { total += hist[(size_t)x]; cum[(size_t)x] = total; }
    std::vector<int> starts{0};
    for (int r = 1; r < world; r++) {
        // first plane whose cumulative count reaches total * r / world (double, as numpy's searchsorted sees it)
        const double want = (double)total * r / world;
        int cut = 0;
        while (cut < gx && (double)cum[(size_t)cut] < want) cut++;
        cut += 1;
        cut = std::max(cut, starts.back() + 2);
        cut = std::min(cut, gx - 2 * (world - r));
        starts.push_back(cut);
    }
    starts.push_back(gx);
    return starts;
}

// The balanced cut, approached from the current one: boundary r moves toward its balanced place only as
// far as at most `budget` particles change hands through it, and no slab becomes wider than max_nx[r] or
// narrower than 2 planes (flow_amd/slab.py: limited_plane_starts is the same rule).
inline std::vector<int> limited_plane_starts(const std::vector<int64_t> &hist, const std::vector<int> &current,
                                             const std::vector<int> &max_nx, int64_t budget)
{
    const int world = (int)current.size() - 1;
    const std::vector<int> want = balanced_plane_starts(hist, world);
    std::vector<int> next = current;
    for (int r = 1; r < world; r++) {
        int cut = current[(size_t)r];
        const int dir = want[(size_t)r] > cut ? 1 : -1;
        int64_t moved = 0;
        while (cut != want[(size_t)r]) {
            const int plane = dir > 0 ? cut : cut - 1;          // the plane that changes hands
            if (moved > 0 && moved + hist[(size_t)plane] > budget) break;      // (one plane always may move)
            moved += hist[(size_t)plane];
            cut += dir;
        }
        next[(size_t)r] = cut;
    }
    // widths: at least 2 planes, at most what the handle was created for (left to right, then right to left)
    for (int r = 1; r < world; r++) next[(size_t)r] = std::max(next[(size_t)r], next[(size_t)r - 1] + 2);
    for (int r = world - 1; r >= 1; r--) next[(size_t)r] = std::min(next[(size_t)r], next[(size_t)r + 1] - 2);
    for (int r = 0; r < world; r++)
        if (next[(size_t)r + 1] - next[(size_t)r] > max_nx[(size_t)r]) return current;      // does not fit: leave the cut alone
    return next;
}

struct MigrationScratch {
    // of the handle's binning, for fill_holes_kernel (null on a handle without binning)
    const int *riding = nullptr;
    int *key = nullptr, *rank = nullptr;
    int *counters = nullptr;      // kCounterWords device words
    int *lists = nullptr;         // 5 device arrays of list_cap ints
    long long list_cap = 0;
    RankTable table{};            // of the classification in flight
    long long classified = -1;    // resident count it ran on (-1: none in flight)
    long long reserve = 4096;     // list entries allocated at the first classification
    long long allocations = 0;    // hipMalloc calls made for the counters and lists
};

inline int migration_reserve(MigrationScratch &M, long long want, std::string *msg)
{
    hipError_t e;
    if (!M.counters) {
        if ((e = hipMalloc(reinterpret_cast<void **>(&M.counters), kCounterWords * sizeof(int))) != hipSuccess) {
            *msg = std::string("hipMalloc counters: ") + hipGetErrorString(e);
            return 1;
        }
        M.allocations += 1;
        if ((e = hipMemset(M.counters, 0, kCounterWords * sizeof(int))) != hipSuccess) {
            *msg = std::string("hipMemset counters: ") + hipGetErrorString(e);
            return 1;
        }
    }
    if (M.list_cap < want) {
        if (M.lists) (void)hipFree(M.lists);
        M.lists = nullptr;
        M.list_cap = 0;
        if ((e = hipMalloc(reinterpret_cast<void **>(&M.lists), (size_t)want * 5 * sizeof(int))) != hipSuccess) {
            *msg = std::string("hipMalloc leaver lists: ") + hipGetErrorString(e);
            return 1;
        }
        M.allocations += 1;
        M.list_cap = want;
    }
    return 0;
}

// Workgroups of find_leavers_kernel: enough to fill the chip, few enough that the counters see few atomics.
inline unsigned find_blocks(long long n) { return (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096)); }

// Enqueues the classification of the n resident particles: leaver counts per destination
// rank and the leaver list.  Nothing is read back here.  dev_counts (may be null) receives
// the n_ranks counts as 64-bit words.  only_if_moved: a classification of the same particles is
// in flight and is repeated only if the store has been re-binned since (`epoch`, a device word that
// counts the re-bins, null for a store without binning); the device decides.
// Returns 0 ok, 1 HIP failure, 3 call order.
inline int find_leavers(MigrationScratch &M, StoreSel st, long long n, int x0, int nx, const int32_t *plane_starts,
                        int n_ranks, long long *dev_counts, const int *epoch, bool only_if_moved,
                        hipStream_t stream, std::string *msg)
{
    if (only_if_moved && (M.classified != n || M.table.n != n_ranks)) {
        *msg = "no classification of these particles is in flight";
        return 3;
    }
    // sized once (MigrationScratch::reserve); pack_leavers grows it when a step has more leavers than that
    if (migration_reserve(M, std::max<long long>(M.list_cap, M.reserve), msg)) return 1;
    M.table = RankTable{};
    M.table.n = n_ranks;
    for (int r = 0; r <= n_ranks; r++) M.table.starts[r] = plane_starts[r];
    hipLaunchKernelGGL(find_prologue_kernel, dim3(1), dim3(128), 0, stream, M.counters, epoch, only_if_moved ? 1 : 0);
    const long long cap = M.list_cap;
    if (n > 0)
        hipLaunchKernelGGL(find_leavers_kernel, dim3(find_blocks(n)), dim3(256), 0, stream, st, n, x0, nx,
                           M.table, M.counters, M.lists, M.lists + cap, M.lists + 2 * cap, cap);
    if (dev_counts)
        hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kMaxRanks), 0, stream, M.counters, n_ranks, dev_counts);
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { *msg = std::string("find_leavers: ") + hipGetErrorString(e); return 1; }
    M.classified = n;
    return 0;
}

// Second half: the caller knows this rank's leaver counts per destination (it read them, or
// the count matrix they are a row of, from the device).  Packs the leavers grouped by
// destination into dev_out and fills their slots from the tail.  Returns 0 ok, 1 HIP failure,
// 2 capacity, 3 call order.
inline int pack_leavers(MigrationScratch &M, StoreSel st, long long n, int x0, int nx, const int64_t *counts,
                        int n_ranks, float *dev_out, long long capacity_records, hipStream_t stream,
                        long long *new_resident, std::string *msg)
{
    *new_resident = n;
    if (M.classified != n || M.table.n != n_ranks) { *msg = "pack_leavers without a matching find_leavers"; return 3; }
    M.classified = -1;
    long long total = 0;
    for (int r = 0; r < n_ranks; r++) total += counts[r];
    if (total == 0) return 0;
    if (total > capacity_records) { *msg = "leaver buffer too small"; return 2; }
    hipError_t e;
    if (total > M.list_cap) {
        // the list overflowed: grow it and classify again (positions have not changed since)
        // (with headroom: growing means freeing, which waits for the device)
        if (migration_reserve(M, 2 * total, msg)) return 1;
        hipLaunchKernelGGL(find_prologue_kernel, dim3(1), dim3(128), 0, stream, M.counters,
                           static_cast<const int *>(nullptr), 0);
        const long long cap = M.list_cap;
        hipLaunchKernelGGL(find_leavers_kernel, dim3(find_blocks(n)), dim3(256), 0, stream, st, n, x0, nx,
                           M.table, M.counters, M.lists, M.lists + cap, M.lists + 2 * cap, cap);
    }
    const long long cap = M.list_cap;
    int *list_slot = M.lists, *list_dest = list_slot + cap, *list_k = list_dest + cap;
    int *holes = list_k + cap, *movers = holes + cap;
    RankOffsets ro{};
    long long acc = 0;
    for (int r = 0; r < n_ranks; r++) { ro.off[r] = acc; acc += counts[r]; }
    const unsigned tb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(pack_leavers_kernel, dim3(tb), dim3(256), 0, stream, st, list_slot, list_dest, list_k,
                       (int)total, ro, dev_out);
    const long long new_n = n - total;
    hipLaunchKernelGGL(find_holes_kernel, dim3(tb), dim3(256), 0, stream, list_slot, (int)total, new_n, M.counters, holes);
    hipLaunchKernelGGL(find_movers_kernel, dim3(tb), dim3(256), 0, stream, st, new_n, n, x0, nx, M.counters, movers);
    hipLaunchKernelGGL(fill_holes_kernel, dim3(tb), dim3(256), 0, stream, st, holes, movers, M.counters, M.riding,
                       M.key, M.rank);
    if ((e = hipGetLastError()) != hipSuccess) { *msg = std::string("compact: ") + hipGetErrorString(e); return 1; }
    *new_resident = new_n;
    return 0;
}

// Both halves with one host synchronisation in between (to learn the leaver counts).
inline int extract_leavers(MigrationScratch &M, StoreSel st, long long n, int x0, int nx, const int32_t *plane_starts,
                           int n_ranks, float *dev_out, long long capacity_records, int64_t *counts,
                           hipStream_t stream, long long *new_resident, std::string *msg)
{
    for (int r = 0; r < n_ranks; r++) counts[r] = 0;
    *new_resident = n;
    if (n == 0) return 0;
    if (find_leavers(M, st, n, x0, nx, plane_starts, n_ranks, nullptr, nullptr, false, stream, msg)) return 1;
    int host_counters[kMaxRanks + 3];
    hipError_t e = hipMemcpyAsync(host_counters, M.counters, sizeof host_counters, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { *msg = std::string("read counters: ") + hipGetErrorString(e); return 1; }
    for (int r = 0; r < n_ranks; r++) counts[r] = host_counters[r];
    return pack_leavers(M, st, n, x0, nx, counts, n_ranks, dev_out, capacity_records, stream, new_resident, msg);
}

}  // namespace mpm
