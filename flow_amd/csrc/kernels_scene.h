// Initial particle states generated on the device, and two whole-store reductions.
//
// The reference fills its particle buffer in a host loop (SIM:114-129: position rejection-sampled in
// the unit ball with Math.random and mapped to ((u * 0.8 + 1) / 2) * gridSize, mass = 1 - rand * 0.002,
// colour = palette[i % 13], everything else zero) and three.js uploads it.  The hosts of this library
// restate that loop with a seeded counter-based generator (flow_amd/particles.py,
// host_js/particles.js); at 2^27 particles the records are 15 GB of host memory, so the same generator
// exists here: every value is computed in binary64 with the host code's operations in the host code's
// order and rounded to binary32 once, so the particles are bit-identical to the host's.
#pragma once

#include "kernels_scatter.h"

namespace mpm {

constexpr int kSceneEllipsoid = 0, kSceneDamBreak = 1;

struct SceneSpec {
    int kind;
    unsigned seed;
    long long first_id, count;
    double gx, gy, gz;          // gridSize (SIM:76)
    double fill;                // dam break: the column spans x in [3, fill * gx]
    float palette[13][3];       // SIM:90-112
    // append mode (slab handles): only particles with floor(x) in [keep_lo, keep_hi) are stored
    int keep_lo, keep_hi;
};

__device__ __forceinline__ unsigned scene_hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// Draw number `draw` of particle `index`: a double in [0, 1) from a 32-bit hash (counter-based, so any
// subset of the particles can be generated on its own).
__device__ __forceinline__ double scene_uniform01(unsigned seed, unsigned index, unsigned draw)
{
    unsigned h = scene_hash32(index ^ seed);
    h = scene_hash32(h + draw * 0x9E3779B9u);
    return (double)h / 4294967296.0;
}

// Position of particle `id`, rounded to binary32 once.
__device__ __forceinline__ void scene_position(const SceneSpec &sp, unsigned id, float *px, float *py, float *pz)
{
    double x, y, z;
    if (sp.kind == kSceneEllipsoid) {
        double ux = 0.0, uy = 0.0, uz = 0.0;
        for (unsigned attempt = 0; attempt < 4096u; attempt++) {
            const double cx = scene_uniform01(sp.seed, id, 3u * attempt + 0u) * 2.0 - 1.0;
            const double cy = scene_uniform01(sp.seed, id, 3u * attempt + 1u) * 2.0 - 1.0;
            const double cz = scene_uniform01(sp.seed, id, 3u * attempt + 2u) * 2.0 - 1.0;
            if (sqrt((cx * cx + cy * cy) + cz * cz) <= 1.0) { ux = cx; uy = cy; uz = cz; break; }
        }
        x = (ux * 0.8 + 1.0) / 2.0 * sp.gx;
        y = (uy * 0.8 + 1.0) / 2.0 * sp.gy;
        z = (uz * 0.8 + 1.0) / 2.0 * sp.gz;
    } else {
        const double hx = fmax(sp.fill * sp.gx, 4.0), hy = sp.gy - 3.0, hz = sp.gz - 3.0;
        x = 3.0 + scene_uniform01(sp.seed, id, 0u) * (hx - 3.0);
        y = 3.0 + scene_uniform01(sp.seed, id, 1u) * (hy - 3.0);
        z = 3.0 + scene_uniform01(sp.seed, id, 2u) * (hz - 3.0);
    }
    *px = (float)x; *py = (float)y; *pz = (float)z;
}

// counter == nullptr: particle first_id + i goes to slot first_slot + i.  Otherwise the kept particles
// are appended from first_slot on in no particular order (one atomic per wave), and *counter ends up as
// their number -- also when they do not all fit below `capacity` (the host checks).
__global__ void __launch_bounds__(256)
generate_scene_kernel(SceneSpec sp, StoreSel ss, long long first_slot, long long capacity, int *counter)
{
    const Store st = ss.pick();
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    // every lane of a wave takes every trip: the ballots below need the whole wave
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < sp.count; i0 += stride) {
        const long long i = i0 + threadIdx.x;
        const bool live = i < sp.count;
        const unsigned id = (unsigned)(sp.first_id + (live ? i : 0));
        float px, py, pz;
        scene_position(sp, id, &px, &py, &pz);
        const float mass = (float)(1.0 - scene_uniform01(sp.seed ^ 0x5BD1E995u, id, 0u) * 0.002);
        long long slot = first_slot + i;
        bool keep = live;
        if (counter) {
            const int ix = f32_to_i32(px);
            keep = live && ix >= sp.keep_lo && ix < sp.keep_hi;
            const unsigned long long kept = __ballot(keep);
            int base = 0;
            if (kept != 0 && lane == __ffsll((long long)kept) - 1) base = atomicAdd(counter, __popcll(kept));
            base = __shfl(base, kept ? __ffsll((long long)kept) - 1 : 0);
            slot = first_slot + base + __popcll(kept & ((1ULL << lane) - 1ULL));
        }
        if (!keep || slot >= capacity) continue;
        const unsigned j = (unsigned)slot;
        st.slot(kPosX)[j] = px; st.slot(kPosY)[j] = py; st.slot(kPosZ)[j] = pz;
#pragma unroll
        for (int k = kVelX; k < kMass; k++) st.slot(k)[j] = 0.0f;          // velocity, C, density
        st.slot(kMass)[j] = mass;
        st.slot(kDirX)[j] = 0.0f; st.slot(kDirY)[j] = 0.0f; st.slot(kDirZ)[j] = 0.0f;
        const int c = (int)(id % 13u);
        st.slot(kColR)[j] = sp.palette[c][0]; st.slot(kColG)[j] = sp.palette[c][1]; st.slot(kColB)[j] = sp.palette[c][2];
        st.ids()[j] = (int)id;
    }
}

// Particles per x-plane of the GLOBAL grid: out[floor(x)] += 1 over slots [0, n).  A workgroup counts in
// LDS (planes <= kHistPlanes) and adds its non-zero words to the 64-bit totals.
constexpr int kHistPlanes = 8192;

__global__ void __launch_bounds__(256)
plane_histogram_kernel(StoreSel ss, long long n, int planes, unsigned long long *__restrict__ out)
{
    __shared__ int local[kHistPlanes];
    const Store st = ss.pick();
    for (int p = threadIdx.x; p < planes; p += blockDim.x) local[p] = 0;
    __syncthreads();
    const long long per = (((n + gridDim.x - 1) / gridDim.x) + 255) / 256 * 256;
    const long long begin = (long long)blockIdx.x * per, end = n < begin + per ? n : begin + per;
    for (long long j = begin + threadIdx.x; j < end; j += blockDim.x) {
        const int ix = min(max(f32_to_i32(st.slot(kPosX)[(unsigned)j]), 0), planes - 1);
        atomicAdd(&local[ix], 1);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < planes; p += blockDim.x)
        if (local[p]) atomicAdd(out + p, (unsigned long long)local[p]);
}

// The same histogram of a scene's particles, straight from the generator (no store): what a host cuts
// balanced slabs from before any slab handle exists.
__global__ void __launch_bounds__(256)
scene_histogram_kernel(SceneSpec sp, int planes, unsigned long long *__restrict__ out)
{
    __shared__ int local[kHistPlanes];
    for (int p = threadIdx.x; p < planes; p += blockDim.x) local[p] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < sp.count; i += stride) {
        float px, py, pz;
        scene_position(sp, (unsigned)(sp.first_id + i), &px, &py, &pz);
        atomicAdd(&local[min(max(f32_to_i32(px), 0), planes - 1)], 1);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < planes; p += blockDim.x)
        if (local[p]) atomicAdd(out + p, (unsigned long long)local[p]);
}

// The fixed-point mass the particles in slots [0, n) deposit in p2g1 (SIM:329: encode(weight) into each
// of the 27 stencil cells), summed in 64 bits straight from the positions -- no grid involved, so it
// checks the grid's mass sum from the other side.  *out must be zero on entry.
__global__ void __launch_bounds__(256)
stencil_mass_sum_kernel(StoreSel ss, long long n, unsigned long long *__restrict__ out)
{
    const Store st = ss.pick();
    long long acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const unsigned q = (unsigned)j;
        const Stencil s = make_stencil(st.slot(kPosX)[q], st.slot(kPosY)[q], st.slot(kPosZ)[q]);
#pragma unroll
        for (int gx = 0; gx < 3; gx++)
#pragma unroll
            for (int gy = 0; gy < 3; gy++) {
                const float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
                for (int gz = 0; gz < 3; gz++) acc += encode_fixed(wxy * s.wz[gz]);
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0 && acc != 0) atomicAdd(out, (unsigned long long)acc);
}

}  // namespace mpm
