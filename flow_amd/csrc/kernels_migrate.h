// Particle migration between x-slabs (no reference counterpart: the reference
// runs on one GPU).  After g2p a particle whose floor(position.x) left the slab's
// owned planes is packed into a 24-word record, grouped by destination rank,
// and removed from the store; the holes are filled from the tail so the
// resident particles stay dense in slots [0, resident).
#pragma once

#include <string>

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

// counters[0..kMaxRanks): leavers per destination; counters[kMaxRanks]: total.
__global__ void __launch_bounds__(256)
find_leavers_kernel(StoreSel ss, long long n, int x0, int nx, RankTable rt, int *counters,
                    int *list_slot, int *list_dest, int *list_k, long long list_cap)
{
    const Store st = ss.pick();
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float px = st.slot(kPosX)[j];
    if (!is_leaver(px, x0, nx)) return;
    int ix = f32_to_i32(px);
    int dest = 0;
    for (int r = 0; r < rt.n; r++)
        if (ix >= rt.starts[r]) dest = r;
    int k = atomicAdd(&counters[dest], 1);
    int e = atomicAdd(&counters[kMaxRanks], 1);
    if (e < list_cap) { list_slot[e] = (int)j; list_dest[e] = dest; list_k[e] = k; }
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

__global__ void __launch_bounds__(256)
fill_holes_kernel(StoreSel ss, const int *holes, const int *movers, const int *counters)
{
    const Store st = ss.pick();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n_holes = counters[kMaxRanks + 1];
    if (i >= n_holes) return;
    long long dst = holes[i], src = movers[i];
#pragma unroll
    for (int w = 0; w < kRecordWords; w++) st.slot(w)[dst] = st.slot(w)[src];
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

// Returns 0 ok, 1 HIP failure, 2 capacity.  One host synchronisation (to learn the
// leaver counts) per call.
inline int extract_leavers(StoreSel st, long long n, const Params &P, int x0, int nx,
                           const int32_t *plane_starts, int n_ranks, float *dev_out,
                           long long capacity_records, int64_t *counts, int **counters_io,
                           int **lists_io, long long *list_cap_io, hipStream_t stream,
                           long long *new_resident, std::string *msg)
{
    (void)P;
    auto hip_fail = [&](const char *what, hipError_t e) {
        *msg = std::string(what) + ": " + hipGetErrorString(e);
        return 1;
    };
    hipError_t e;
    for (int r = 0; r < n_ranks; r++) counts[r] = 0;
    *new_resident = n;
    if (n == 0) return 0;
    const int n_counters = kMaxRanks + 3;
    if (!*counters_io) {
        if ((e = hipMalloc(reinterpret_cast<void **>(counters_io), n_counters * sizeof(int))) != hipSuccess)
            return hip_fail("hipMalloc counters", e);
    }
    RankTable rt{};
    rt.n = n_ranks;
    for (int r = 0; r <= n_ranks; r++) rt.starts[r] = plane_starts[r];
    int host_counters[n_counters];
    long long want = std::max<long long>(*list_cap_io, 4096);
    for (int attempt = 0; attempt < 2; attempt++) {
        if (*list_cap_io < want) {
            if (*lists_io) (void)hipFree(*lists_io);
            *lists_io = nullptr;
            if ((e = hipMalloc(reinterpret_cast<void **>(lists_io), (size_t)want * 5 * sizeof(int))) != hipSuccess)
                return hip_fail("hipMalloc leaver lists", e);
            *list_cap_io = want;
        }
        const long long cap = *list_cap_io;
        int *list_slot = *lists_io, *list_dest = list_slot + cap, *list_k = list_dest + cap;
        if ((e = hipMemsetAsync(*counters_io, 0, n_counters * sizeof(int), stream)) != hipSuccess)
            return hip_fail("memset counters", e);
        hipLaunchKernelGGL(find_leavers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, st, n,
                           x0, nx, rt, *counters_io, list_slot, list_dest, list_k, cap);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail("find_leavers", e);
        if ((e = hipMemcpyAsync(host_counters, *counters_io, n_counters * sizeof(int), hipMemcpyDeviceToHost,
                                stream)) != hipSuccess)
            return hip_fail("copy counters", e);
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return hip_fail("sync", e);
        if (host_counters[kMaxRanks] <= cap) break;
        want = host_counters[kMaxRanks];       // list too small: grow and classify again
    }
    const int total = host_counters[kMaxRanks];
    if (total == 0) return 0;
    if (total > capacity_records) {
        *msg = "leaver buffer too small";
        return 2;
    }
    const long long cap = *list_cap_io;
    int *list_slot = *lists_io, *list_dest = list_slot + cap, *list_k = list_dest + cap;
    int *holes = list_k + cap, *movers = holes + cap;
    RankOffsets ro{};
    long long acc = 0;
    for (int r = 0; r < n_ranks; r++) {
        ro.off[r] = acc;
        counts[r] = host_counters[r];
        acc += host_counters[r];
    }
    const unsigned tb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(pack_leavers_kernel, dim3(tb), dim3(256), 0, stream, st, list_slot, list_dest, list_k,
                       total, ro, dev_out);
    const long long new_n = n - total;
    hipLaunchKernelGGL(find_holes_kernel, dim3(tb), dim3(256), 0, stream, list_slot, total, new_n, *counters_io,
                       holes);
    hipLaunchKernelGGL(find_movers_kernel, dim3(tb), dim3(256), 0, stream, st, new_n, n, x0, nx, *counters_io,
                       movers);
    hipLaunchKernelGGL(fill_holes_kernel, dim3(tb), dim3(256), 0, stream, st, holes, movers, *counters_io);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail("compact", e);
    *new_resident = new_n;
    return 0;
}

}  // namespace mpm
