// The reference's own algorithm on CDNA4: one thread per particle, every stencil
// contribution added to the global fixed-point grid with an i32 atomic
// (SIM:315-331, 377-391), and one thread per cell for the two grid sweeps.
// This is the baseline path (FLOWMPM_ALGO_SCATTER); the tiled path shares the
// grid sweeps and the arithmetic in mpm_device.h.
#pragma once

#include "mpm_device.h"
#include "mpm_modes.h"

namespace mpm {

// One slot array of the store, indexed by particle slot.  The byte offset is formed in 32 bits
// (capacity is limited to 2^30 elements for that), so an access is "uniform 64-bit base +
// 32-bit lane offset": the global_load / global_store form with a scalar base register, and
// no 64-bit address arithmetic per lane and access.
template <typename T>
struct Column {
    T *p;
    __device__ __forceinline__ T &operator[](unsigned i) const
    {
        return *reinterpret_cast<T *>(reinterpret_cast<char *>(p) + (size_t)(i << 2));
    }
    __device__ __forceinline__ T *operator+(unsigned i) const { return &(*this)[i]; }
};

constexpr long long kMaxCapacity = 1LL << 30;

struct Store {
    float *base;        // kNumSlots arrays of `capacity` elements, slot-major
    long long capacity;
    __device__ __forceinline__ Column<float> slot(int k) const { return Column<float>{base + (long long)k * capacity}; }
    __device__ __forceinline__ Column<int> ids() const
    {
        return Column<int>{reinterpret_cast<int *>(base + (long long)kId * capacity)};
    }
};

// The two particle stores of a handle and the device word that says which one is current.
// Re-binning gathers the current store into the other one and flips the word on the device, so
// the host can enqueue steps without knowing whether a re-bin took place.  cur == nullptr:
// a handle with a single store (the scatter algorithm).
struct StoreSel {
    float *base[2];
    long long capacity;
    const int *cur;
    __device__ __forceinline__ Store pick() const { return Store{(cur && *cur) ? base[1] : base[0], capacity}; }
    __device__ __forceinline__ Store other() const { return Store{(cur && *cur) ? base[0] : base[1], capacity}; }
};

// The fixed-point grid (cellBuffer, SIM:131-138) as the device holds it: momentum in 16-byte
// records {x, y, z, unused} and the mass words in an array of their own, so that the mass
// pass of the tiled algorithm flushes contiguous 4-byte atomics (a 64-byte memory-side atomic
// request then carries 16 cells instead of 4) and the density gather reads 4 bytes per cell.
// Hosts always see the reference's {x, y, z, mass} records: downloads and halo planes are
// assembled by kernels below.
struct FixedGrid {
    int *cell;      // 4 ints per cell; word 3 is not used on the device
    int *mass;      // 1 int per cell
    // Tiled handles only (else null): one word per 8x8x8 tile of cells whose bits say which
    // buffers may hold a non-zero word inside the tile -- bit 0 the momentum records, bits 1 and
    // 2 the two mass buffers.  Every kernel that adds to the grid sets `dirty_bits` for the tiles
    // it touches; clearGrid and updateGrid skip the tiles whose bits are clear.
    int *dirty;
    int dirty_bits;
};

// Local linear cell index (SIM:286-290) or -1 when the cell is outside the planes
// this handle holds (only reachable from positions the host refused to upload).
__device__ __forceinline__ long long local_cell(int cx, int cy, int cz, const Params &P)
{
    int lx = cx - P.x_off;
    if ((unsigned)lx >= (unsigned)P.local_nx || (unsigned)cy >= (unsigned)P.gy ||
        (unsigned)cz >= (unsigned)P.gz)
        return -1;
    return ((long long)lx * P.gy + cy) * P.gz + cz;
}

// SIM:262-277.  One thread clears one cell.  FLOAT_GRID = true clears both grids as the
// reference does (and the mass words that describe the float grid, see update_grid_kernel);
// FLOAT_GRID = false clears the fixed-point grid only and leaves the float grid to
// update_grid_kernel: 20 instead of 36 bytes per cell.
template <bool FLOAT_GRID>
__global__ void __launch_bounds__(256)
clear_grid_kernel(int4 *__restrict__ cell, int *__restrict__ mass, float4 *__restrict__ cellf,
                  int *__restrict__ mass_of_cellf, long long cells)
{
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
        cell[i] = make_int4(0, 0, 0, 0);
        mass[i] = 0;
        if (FLOAT_GRID) {
            cellf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            mass_of_cellf[i] = 0;
        }
    }
}

// SIM:295-333.
__global__ void __launch_bounds__(256)
p2g1_scatter_kernel(StoreSel ss, long long n, FixedGrid g, Params P)
{
    const Store st = ss.pick();
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float px = st.slot(kPosX)[j], py = st.slot(kPosY)[j], pz = st.slot(kPosZ)[j];
    float vx = st.slot(kVelX)[j], vy = st.slot(kVelY)[j], vz = st.slot(kVelZ)[j];
    float C[9];
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = st.slot(kC00 + k)[j];
    Stencil s = make_stencil(px, py, pz);
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
                long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                if (c < 0) continue;
                int *p = g.cell + 4 * c;
                atomicAdd(p + 0, encode_fixed(weight * (vx + qx)));
                atomicAdd(p + 1, encode_fixed(weight * (vy + qy)));
                atomicAdd(p + 2, encode_fixed(weight * (vz + qz)));
                atomicAdd(g.mass + c, encode_fixed(weight));
            }
        }
    }
}

// SIM:336-392.
__global__ void __launch_bounds__(256)
p2g2_scatter_kernel(StoreSel ss, long long n, FixedGrid g, Params P)
{
    const Store st = ss.pick();
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float px = st.slot(kPosX)[j], py = st.slot(kPosY)[j], pz = st.slot(kPosZ)[j];
    float C[9];
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = st.slot(kC00 + k)[j];
    Stencil s = make_stencil(px, py, pz);

    float density = 0.0f;
#pragma unroll
    for (int gx = 0; gx < 3; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                int m = (c >= 0) ? g.mass[c] : 0;
                density = density + decode_fixed(m) * weight;
            }
        }
    const Column<float> dens = st.slot(kDensity);
    dens[j] = dens[j] * (1.0f - 0.05f) + density * 0.05f;      // SIM:365-366

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
                float mx = (a00 * dx + a01 * dy) + a02 * dz;
                float my = (a01 * dx + a11 * dy) + a12 * dz;
                float mz = (a02 * dx + a12 * dy) + a22 * dz;
                long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                if (c < 0) continue;
                int *p = g.cell + 4 * c;
                atomicAdd(p + 0, encode_fixed(mx));
                atomicAdd(p + 1, encode_fixed(my));
                atomicAdd(p + 2, encode_fixed(mz));
            }
        }
}

// SIM:395-429.  Cells with mass <= 0 keep what clearGrid wrote (zeros).  The float grid is
// non-zero only where `mass_of_cellf` -- the mass words of the step that last wrote it -- is
// positive, so when clearGrid skipped the float grid the zeros are written here, and only into
// the cells the fluid has left since.  mass_of_cellf may be mass_i itself (float grid cleared).
__device__ __forceinline__ void update_grid_cell(const int4 *__restrict__ cell, const int *__restrict__ mass_i,
                                                 const int *__restrict__ mass_of_cellf, float4 *__restrict__ cellf,
                                                 long long i, int x, int y, int z, const Params &P)
{
    const int before = mass_of_cellf[i];
    float mass = decode_fixed(mass_i[i]);
    if (!(mass > 0.0f)) {
        if (before > 0) cellf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    int4 c = cell[i];
    float vx = decode_fixed(c.x) / mass;
    float vy = decode_fixed(c.y) / mass;
    float vz = decode_fixed(c.z) / mass;
    if (x < 2 || x > P.gx - 2) vx = 0.0f;
    if (y < 2 || y > P.gy - 2) vy = 0.0f;
    if (z < 2 || z > P.gz - 2) vz = 0.0f;
    cellf[i] = make_float4(vx, vy, vz, mass);
}

// The same with the three words of the cell already loaded (c is unused when the cell holds no mass).
__device__ __forceinline__ void update_grid_cell_loaded(int4 c, int mass_word, int before, float4 *__restrict__ cellf,
                                                        long long i, int x, int y, int z, const Params &P)
{
    float mass = decode_fixed(mass_word);
    if (!(mass > 0.0f)) {
        if (before > 0) cellf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float vx = decode_fixed(c.x) / mass;
    float vy = decode_fixed(c.y) / mass;
    float vz = decode_fixed(c.z) / mass;
    if (x < 2 || x > P.gx - 2) vx = 0.0f;
    if (y < 2 || y > P.gy - 2) vy = 0.0f;
    if (z < 2 || z > P.gz - 2) vz = 0.0f;
    cellf[i] = make_float4(vx, vy, vz, mass);
}

__global__ void __launch_bounds__(256)
update_grid_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass_i,
                   const int *__restrict__ mass_of_cellf, float4 *__restrict__ cellf, long long cells, Params P)
{
    long long stride = (long long)gridDim.x * blockDim.x;
    const int plane = P.gy * P.gz;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
        int lx = (int)(i / plane);
        int rem = (int)(i - (long long)lx * plane);
        int y = rem / P.gz;
        update_grid_cell(cell, mass_i, mass_of_cellf, cellf, i, lx + P.x_off, y, rem - y * P.gz, P);
    }
}

// The same over two runs of `cells` cells each, starting at first[0] and first[1] (a slab's two pairs of boundary
// planes after the neighbours' momentum has been merged into them).
__global__ void __launch_bounds__(256)
update_grid_ranges_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass_i,
                          const int *__restrict__ mass_of_cellf, float4 *__restrict__ cellf, long long first0,
                          long long first1, long long cells, Params P)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int plane = P.gy * P.gz;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < 2 * cells; k += stride) {
        const long long i = k < cells ? first0 + k : first1 + (k - cells);
        int lx = (int)(i / plane);
        int rem = (int)(i - (long long)lx * plane);
        int y = rem / P.gz;
        update_grid_cell(cell, mass_i, mass_of_cellf, cellf, i, lx + P.x_off, y, rem - y * P.gz, P);
    }
}

// SIM:431-1536, gathering straight from the global float grid.  MODES = false is the chaos
// path (every mode uniform zero, M unused); MODES = true adds the mode branches.
template <bool MODES>
__global__ void __launch_bounds__(256)
g2p_gather_kernel(StoreSel ss, long long n, const float4 *__restrict__ cellf, Params P,
                  const ModeParams *__restrict__ M)
{
    const Store st = ss.pick();
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    G2PIn in;
    in.px = st.slot(kPosX)[j]; in.py = st.slot(kPosY)[j]; in.pz = st.slot(kPosZ)[j];
    in.mass = st.slot(kMass)[j];
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
    const bool gather = !MODES || (M->grid_mode == 0u && !hide);   // SIM:1332
#pragma unroll
    for (int gx = 0; gx < 3 && gather; gx++)
#pragma unroll
        for (int gy = 0; gy < 3; gy++) {
            float wxy = s.wx[gx] * s.wy[gy];
#pragma unroll
            for (int gz = 0; gz < 3; gz++) {
                float weight = wxy * s.wz[gz];
                long long c = local_cell(s.cx + gx, s.cy + gy, s.cz + gz, P);
                float4 f = (c >= 0) ? cellf[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float wvx = f.x * weight, wvy = f.y * weight, wvz = f.z * weight;
                float dx = s.dx[gx], dy = s.dy[gy], dz = s.dz[gz];
                B[0] = B[0] + wvx * dx; B[1] = B[1] + wvy * dx; B[2] = B[2] + wvz * dx;
                B[3] = B[3] + wvx * dy; B[4] = B[4] + wvy * dy; B[5] = B[5] + wvz * dy;
                B[6] = B[6] + wvx * dz; B[7] = B[7] + wvy * dz; B[8] = B[8] + wvz * dz;
                v.x = v.x + wvx; v.y = v.y + wvy; v.z = v.z + wvz;
            }
        }
    G2POut o = MODES ? g2p_finish_modes(in, v, B, index, hide, P, M) : g2p_finish(in, v, B, P);
#pragma unroll
    for (int k = 0; k < 9; k++) st.slot(kC00 + k)[j] = o.C[k];
    st.slot(kPosX)[j] = o.px; st.slot(kPosY)[j] = o.py; st.slot(kPosZ)[j] = o.pz;
    st.slot(kVelX)[j] = o.vx; st.slot(kVelY)[j] = o.vy; st.slot(kVelZ)[j] = o.vz;
    st.slot(kDirX)[j] = o.dirx; st.slot(kDirY)[j] = o.diry; st.slot(kDirZ)[j] = o.dirz;
}

// Host AoS record (28 floats, the reference's layout) <-> SoA store.  Without `ids`, record i becomes
// particle id_base + i.
__global__ void __launch_bounds__(256)
aos_to_soa_kernel(const float *__restrict__ aos, const int *__restrict__ ids, StoreSel ss,
                  long long first_slot, long long count, long long id_base)
{
    const Store st = ss.pick();
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float *r = aos + i * 28;
    long long j = first_slot + i;
    st.slot(kPosX)[j] = r[0]; st.slot(kPosY)[j] = r[1]; st.slot(kPosZ)[j] = r[2];
    st.slot(kDensity)[j] = r[3];
    st.slot(kVelX)[j] = r[4]; st.slot(kVelY)[j] = r[5]; st.slot(kVelZ)[j] = r[6];
    st.slot(kMass)[j] = r[7];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++) st.slot(kC00 + 3 * c + row)[j] = r[8 + 4 * c + row];
    st.slot(kDirX)[j] = r[20]; st.slot(kDirY)[j] = r[21]; st.slot(kDirZ)[j] = r[22];
    st.slot(kColR)[j] = r[24]; st.slot(kColG)[j] = r[25]; st.slot(kColB)[j] = r[26];
    st.ids()[j] = ids ? ids[i] : (int)(id_base + i);
}

// dst_by_id != 0: record of the particle in slot j goes to row ids[j] (id order); otherwise to row j,
// and ids_out receives its id.  Only rows [row_lo, row_lo + rows) are written, to aos[(row - row_lo) * 28]
// (and ids_out[row - row_lo]): a large store is read back through a staging buffer of fixed size, one
// window of rows at a time.  In id order every window scans all `count` slots; in slot order a window
// visits its own slots only.
__global__ void __launch_bounds__(256)
soa_to_aos_kernel(StoreSel ss, long long count, float *__restrict__ aos, int *__restrict__ ids_out,
                  int dst_by_id, long long row_lo, long long rows)
{
    const Store st = ss.pick();
    long long j = (dst_by_id ? 0 : row_lo) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    int id = st.ids()[j];
    const long long row = (dst_by_id ? (long long)id : j) - row_lo;
    if (row < 0 || row >= rows) return;
    float *r = aos + row * 28;
    r[0] = st.slot(kPosX)[j]; r[1] = st.slot(kPosY)[j]; r[2] = st.slot(kPosZ)[j];
    r[3] = st.slot(kDensity)[j];
    r[4] = st.slot(kVelX)[j]; r[5] = st.slot(kVelY)[j]; r[6] = st.slot(kVelZ)[j];
    r[7] = st.slot(kMass)[j];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int row3 = 0; row3 < 3; row3++) r[8 + 4 * c + row3] = st.slot(kC00 + 3 * c + row3)[j];
        r[8 + 4 * c + 3] = 0.0f;
    }
    r[20] = st.slot(kDirX)[j]; r[21] = st.slot(kDirY)[j]; r[22] = st.slot(kDirZ)[j];
    r[23] = 0.0f;
    r[24] = st.slot(kColR)[j]; r[25] = st.slot(kColG)[j]; r[26] = st.slot(kColB)[j];
    r[27] = 0.0f;
    if (ids_out) ids_out[row] = id;
}

// One per-particle field in id order (what a renderer binds): the values of ids [id_lo, id_lo + ids)
// go to out[(id - id_lo) * width].
__global__ void __launch_bounds__(256)
gather_field_kernel(StoreSel ss, long long count, int first_slot, int width, float *__restrict__ out,
                    long long id_lo, long long ids)
{
    const Store st = ss.pick();
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const long long row = (long long)st.ids()[j] - id_lo;
    if (row < 0 || row >= ids) return;
    for (int k = 0; k < width; k++) out[row * width + k] = st.slot(first_slot + k)[j];
}

// One per-particle field of slots [first, first + rows) in slot order, with the particles' ids beside it
// (a slab's share of a field; the host of a slab group puts the shares in id order).
__global__ void __launch_bounds__(256)
gather_field_slots_kernel(StoreSel ss, long long first, long long rows, int first_slot, int width,
                          float *__restrict__ out, int *__restrict__ ids_out)
{
    const Store st = ss.pick();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const long long j = first + i;
    for (int k = 0; k < width; k++) out[i * width + k] = st.slot(first_slot + k)[j];
    ids_out[i] = st.ids()[j];
}

// Halo merge of the slab decomposition: dst.comp += src.comp (wrapping i32).
__global__ void __launch_bounds__(256)
plane_add_kernel(int4 *__restrict__ dst, int *__restrict__ dst_mass, const int4 *__restrict__ src,
                 long long cells, unsigned mask)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    int4 s = src[i];
    if (mask & 7u) {
        int4 d = dst[i];
        if (mask & 1u) d.x = (int)((unsigned)d.x + (unsigned)s.x);
        if (mask & 2u) d.y = (int)((unsigned)d.y + (unsigned)s.y);
        if (mask & 4u) d.z = (int)((unsigned)d.z + (unsigned)s.z);
        dst[i] = d;
    }
    if (mask & 8u) dst_mass[i] = (int)((unsigned)dst_mass[i] + (unsigned)s.w);
}

// Halo exchange with only the components a merge needs: `cells` consecutive cells (the two
// boundary planes of one side are adjacent in memory) packed as k = popcount(mask) words per
// cell, components in x, y, z, mass order.
__global__ void __launch_bounds__(256)
halo_pack_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass, int *__restrict__ out,
                 long long cells, unsigned mask, int k)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    int *o = out + i * k;
    if (mask & 7u) {
        int4 c = cell[i];
        if (mask & 1u) *o++ = c.x;
        if (mask & 2u) *o++ = c.y;
        if (mask & 4u) *o++ = c.z;
    }
    if (mask & 8u) *o = mass[i];
}

__global__ void __launch_bounds__(256)
halo_add_kernel(int4 *__restrict__ dst, int *__restrict__ dst_mass, const int *__restrict__ src,
                long long cells, unsigned mask, int k)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const int *p = src + i * k;
    if (mask & 7u) {
        int4 d = dst[i];
        if (mask & 1u) d.x = (int)((unsigned)d.x + (unsigned)*p++);
        if (mask & 2u) d.y = (int)((unsigned)d.y + (unsigned)*p++);
        if (mask & 4u) d.z = (int)((unsigned)d.z + (unsigned)*p++);
        dst[i] = d;
    }
    if (mask & 8u) dst_mass[i] = (int)((unsigned)dst_mass[i] + (unsigned)*p);
}

// Both sides of a slab in one launch (blockIdx.y = side): a side without a neighbour has a null buffer.
struct HaloSides {
    long long first[2];      // first cell of the side's plane pair
    int *buf[2];             // packed words of that side (pack: destination, add: source)
};

__global__ void __launch_bounds__(256)
halo_pack_sides_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass, HaloSides h, long long cells,
                       unsigned mask, int k)
{
    const int side = blockIdx.y;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!h.buf[side] || i >= cells) return;
    int *o = h.buf[side] + i * k;
    const long long c = h.first[side] + i;
    if (mask & 7u) {
        int4 v = cell[c];
        if (mask & 1u) *o++ = v.x;
        if (mask & 2u) *o++ = v.y;
        if (mask & 4u) *o++ = v.z;
    }
    if (mask & 8u) *o = mass[c];
}

__global__ void __launch_bounds__(256)
halo_add_sides_kernel(int4 *__restrict__ cell, int *__restrict__ mass, HaloSides h, long long cells, unsigned mask, int k)
{
    const int side = blockIdx.y;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!h.buf[side] || i >= cells) return;
    const int *p = h.buf[side] + i * k;
    const long long c = h.first[side] + i;
    if (mask & 7u) {
        int4 d = cell[c];
        if (mask & 1u) d.x = (int)((unsigned)d.x + (unsigned)*p++);
        if (mask & 2u) d.y = (int)((unsigned)d.y + (unsigned)*p++);
        if (mask & 4u) d.z = (int)((unsigned)d.z + (unsigned)*p++);
        cell[c] = d;
    }
    if (mask & 8u) mass[c] = (int)((unsigned)mass[c] + (unsigned)*p);
}

// 64-bit sums of the four fixed-point components over `cells` consecutive cells (conservation
// tracking: the grid mass is an integer sum that the step conserves exactly).  out[0..3] must
// be zero on entry.
__global__ void __launch_bounds__(256)
grid_sums_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass, long long cells,
                 unsigned long long *__restrict__ out)
{
    long long acc[4] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
        int4 c = cell[i];
        acc[0] += c.x; acc[1] += c.y; acc[2] += c.z; acc[3] += mass[i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(out + k, (unsigned long long)v);
    }
}

// The reference's cell records {x, y, z, mass} of `cells` consecutive cells, assembled from
// the device layout (downloads, halo-plane send staging).
__global__ void __launch_bounds__(256)
assemble_cells_kernel(const int4 *__restrict__ cell, const int *__restrict__ mass, int4 *__restrict__ out,
                      long long cells)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    int4 c = cell[i];
    c.w = mass[i];
    out[i] = c;
}

}  // namespace mpm
