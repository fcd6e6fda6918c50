// Microbenchmark: what the mass pass of p2g spends its time on.  2^24 particles sorted by 8x8x8 tile
// (1,900 per tile, uniformly placed inside it), one workgroup of 256 threads per tile with a 12^3-cell
// LDS arena, as p2g_mass_tiled_kernel; pieces switched off one at a time.  All buffers belong to this
// program.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o mass_pass mass_pass.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kG = 256, kTiles = 32, kA = 12, kCells = kA * kA * kA;

__device__ __forceinline__ void axis(float p, int &c, float w[3])
{
    c = (int)p - 1;
    const float d = (p - floorf(p)) - 0.5f;
    w[0] = (0.5f * (0.5f - d)) * (0.5f - d); w[1] = 0.75f - d * d; w[2] = (0.5f * (0.5f + d)) * (0.5f + d);
}

// MODE bits: 1 position loads (else positions from the index), 2 the 27 LDS atomics, 4 the flush to the global
// mass array, 8 arena zeroing, 16 one atomicOr per touched tile (27 at most per workgroup, read first),
// 32 software prefetch of the next particle's position
template <int MODE>
__global__ void __launch_bounds__(256)
mass_kernel(const float *px, const float *py, const float *pz, int per_tile, int n_items, int *mass, int *dirty)
{
    const int tile = blockIdx.x;
    if (tile >= n_items) return;
    __shared__ int lds[kCells];
    const int tz = tile % kTiles, ty = (tile / kTiles) % kTiles, tx = tile / (kTiles * kTiles);
    const int ox = tx * 8 - 2, oy = ty * 8 - 2, oz = tz * 8 - 2;
    if (MODE & 8) for (int i = threadIdx.x; i < kCells; i += 256) lds[i] = 0;
    __syncthreads();
    int sink = 0;
    const int j_end = (tile + 1) * per_tile;
    float nx = 0, ny = 0, nz = 0;
    if (MODE & 32) { const int j = tile * per_tile + threadIdx.x; if (j < j_end) { nx = px[j]; ny = py[j]; nz = pz[j]; } }
    for (int j = tile * per_tile + threadIdx.x; j < j_end; j += 256) {
        float x, y, z;
        if (MODE & 32) {                      // the next particle's position is loaded before this one's atomics
            x = nx; y = ny; z = nz;
            if (j + 256 < j_end) { nx = px[j + 256]; ny = py[j + 256]; nz = pz[j + 256]; }
        }
        else if (MODE & 1) { x = px[j]; y = py[j]; z = pz[j]; }
        else { x = tx * 8 + (j % 61) * 0.13f; y = ty * 8 + (j % 59) * 0.135f; z = tz * 8 + (j % 53) * 0.15f; }
        int cx, cy, cz; float wx[3], wy[3], wz[3];
        axis(x, cx, wx); axis(y, cy, wy); axis(z, cz, wz);
        const int base = ((cx - ox) * kA + (cy - oy)) * kA + (cz - oz);
#pragma unroll
        for (int gx = 0; gx < 3; gx++)
#pragma unroll
            for (int gy = 0; gy < 3; gy++) {
                const float wxy = wx[gx] * wy[gy];
#pragma unroll
                for (int gz = 0; gz < 3; gz++) {
                    const int em = (int)((wxy * wz[gz]) * 1.0e7f);
                    if (MODE & 2) atomicAdd(&lds[base + (gx * kA + gy) * kA + gz], em); else sink += em;
                }
            }
    }
    __syncthreads();
    if (MODE & 4)
        for (int a = threadIdx.x; a < kCells; a += 256) {
            const int v = (MODE & 2) ? lds[a] : a + 1;
            if (v == 0) continue;
            const int gx = ox + a / (kA * kA), gy = oy + (a / kA) % kA, gz = oz + a % kA;
            if ((unsigned)gx < (unsigned)kG && (unsigned)gy < (unsigned)kG && (unsigned)gz < (unsigned)kG)
                atomicAdd(&mass[(gx * kG + gy) * kG + gz], v);
        }
    if ((MODE & 16) && threadIdx.x < 27) {
        const int k = threadIdx.x, nx = tx + k / 9 - 1, ny = ty + (k / 3) % 3 - 1, nz = tz + k % 3 - 1;
        if ((unsigned)nx < (unsigned)kTiles && (unsigned)ny < (unsigned)kTiles && (unsigned)nz < (unsigned)kTiles) {
            int *word = &dirty[(nx * kTiles + ny) * kTiles + nz];
            if ((*word & 3) != 3) atomicOr(word, 3);
        }
    }
    if (sink == 0x7fffffff) mass[0] = sink;
}

template <int MODE>
static float run(const float *px, const float *py, const float *pz, int per_tile, int n_items, int *mass, int *dirty)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(dirty, 0, kTiles * kTiles * kTiles * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mass_kernel<MODE>, dim3(n_items), dim3(256), 0, 0, px, py, pz, per_tile, n_items, mass, dirty);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int per_tile = 1900, n_items = (1 << 24) / per_tile, n = n_items * per_tile;
    std::vector<float> x(n), y(n), z(n);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int j = 0; j < n; j++) {
        const int t = j / per_tile, tz = t % kTiles, ty = (t / kTiles) % kTiles, tx = t / (kTiles * kTiles);
        x[j] = tx * 8 + 7.99f * rnd(); y[j] = ty * 8 + 7.99f * rnd(); z[j] = tz * 8 + 7.99f * rnd();
        if (x[j] < 2.0f) x[j] = 2.0f; if (y[j] < 2.0f) y[j] = 2.0f; if (z[j] < 2.0f) z[j] = 2.0f;
    }
    float *px, *py, *pz; int *mass, *dirty;
    CHECK(hipMalloc(&px, (size_t)n * 4)); CHECK(hipMalloc(&py, (size_t)n * 4)); CHECK(hipMalloc(&pz, (size_t)n * 4));
    CHECK(hipMalloc(&mass, (size_t)kG * kG * kG * 4)); CHECK(hipMalloc(&dirty, kTiles * kTiles * kTiles * 4));
    CHECK(hipMemset(mass, 0, (size_t)kG * kG * kG * 4));
    CHECK(hipMemcpy(px, x.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(py, y.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pz, z.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    printf("%d work items of %d particles\n", n_items, per_tile);
    printf("  everything                          %.3f ms\n", run<31>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  everything, next position prefetched %.3f ms\n", run<63>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  without the tile marks              %.3f ms\n", run<15>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  without the flush                   %.3f ms\n", run<27>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  without the LDS atomics             %.3f ms\n", run<29>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  without the position loads          %.3f ms\n", run<30>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  loads + LDS atomics + zeroing only  %.3f ms\n", run<11>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  loads + zeroing + flush of all cells %.3f ms\n", run<13>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  loads only                          %.3f ms\n", run<1>(px, py, pz, per_tile, n_items, mass, dirty));
    printf("  LDS atomics only                    %.3f ms\n", run<2>(px, py, pz, per_tile, n_items, mass, dirty));
    return 0;
}
