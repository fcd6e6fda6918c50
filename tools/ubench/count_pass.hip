// Microbenchmark: what the count pass of the re-bin spends its time on.  Synthetic store of n
// particles that is sorted by tile up to a few strays (as the real one is when a re-bin is due);
// the same loop as bin_count_scan_kernel with pieces switched off one at a time.  Everything lives
// in buffers of this program, so the reduced variants can be run safely.
// Build: hipcc -O3 --offload-arch=gfx950 -o count_pass count_pass.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE bits: 1 loads of the three position arrays (else the key is computed from j),
//            2 the returning atomic on the tile counter (else base = 0), 4 stores of key and rank
template <int MODE, bool GRID_STRIDE>
__global__ void __launch_bounds__(256)
count_kernel(const float *px, const float *py, const float *pz, int n, int per_tile, int *tile_count, int *key, int *rank)
{
    const int lane = threadIdx.x & 63;
    long long begin, end, step;
    if (GRID_STRIDE) { begin = (long long)blockIdx.x * 256; end = n; step = (long long)gridDim.x * 256; }
    else {
        const long long per = (((long long)n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
        begin = (long long)blockIdx.x * per; end = begin + per < n ? begin + per : n; step = 256;
    }
    int sink = 0;
    for (long long j0 = begin; j0 < end; j0 += step) {
        const long long j = j0 + threadIdx.x;
        const bool valid = j < end;
        int k = -1 - lane;
        if (valid) {
            if (MODE & 1) k = ((int)px[j] / 8 * 32 + (int)py[j] / 8) * 32 + (int)pz[j] / 8;
            else k = (int)(j / per_tile);
        }
        int base = 0, start = lane;
        if (MODE & 8) {
            // one atomic per distinct key in the wave, wherever its lanes are
            const unsigned long long below = (1ULL << lane) - 1ULL;
            int r = 0;
            for (unsigned long long rest = __ballot(valid); rest != 0;) {
                const int leader = __ffsll((long long)rest) - 1;
                const int kl = __shfl(k, leader);
                const unsigned long long same = __ballot(valid && k == kl);
                int b = 0;
                if ((MODE & 2) && lane == leader) b = atomicAdd(&tile_count[kl], __popcll(same));
                b = __shfl(b, leader);
                if (valid && k == kl) r = b + __popcll(same & below);
                rest &= ~same;
            }
            base = r; start = lane;
        } else {
        int prev = __shfl_up(k, 1);
        bool head = (lane == 0) || (k != prev);
        unsigned long long heads = __ballot(head);
        unsigned long long upto = (2ULL << lane) - 1ULL;
        start = 63 - __clzll((long long)(heads & upto));
        unsigned long long above = heads & ~upto;
        int stop = above ? (__ffsll((long long)above) - 1) : 64;
        if ((MODE & 2) && head && valid) base = atomicAdd(&tile_count[k], stop - start);
        base = __shfl(base, start);
        }
        if (valid) {
            if (MODE & 4) { key[j] = k; rank[j] = base + (lane - start); }
            else sink += k + base;
        }
    }
    if (sink == 0x7fffffff) key[0] = sink;
}

template <int MODE, bool GS>
static float run(const float *px, const float *py, const float *pz, int n, int per_tile, int *tc, int *key, int *rank,
                 int n_tiles, int blocks)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(tc, 0, n_tiles * sizeof(int)));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((count_kernel<MODE, GS>), dim3(blocks), dim3(256), 0, 0, px, py, pz, n, per_tile, tc, key, rank);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int n = 1 << 24, per_tile = 1900, n_tiles = 32 * 32 * 32;
    const int stray_one_in = argc > 1 ? atoi(argv[1]) : 400;     // 1 / fraction of particles that sit in a neighbouring tile
    std::vector<float> x(n), y(n), z(n);
    unsigned s = 12345u;
    for (int j = 0; j < n; j++) {
        int t = j / per_tile;
        s = s * 1664525u + 1013904223u;
        if ((s >> 8) % stray_one_in == 0) {                                   // a stray: one tile further in z, y or x
            const int hop[6] = {1, n_tiles - 1, 32, n_tiles - 32, 1024, n_tiles - 1024};
            t = (t + hop[(s >> 20) % 6]) % n_tiles;
        }
        int tz = t % 32, ty = (t / 32) % 32, tx = t / 1024;
        x[j] = tx * 8 + 0.5f + (j % 7); y[j] = ty * 8 + 0.5f + (j % 5); z[j] = tz * 8 + 0.5f + (j % 3);
    }
    float *px, *py, *pz; int *tc, *key, *rank;
    CHECK(hipMalloc(&px, n * 4)); CHECK(hipMalloc(&py, n * 4)); CHECK(hipMalloc(&pz, n * 4));
    CHECK(hipMalloc(&tc, n_tiles * 4)); CHECK(hipMalloc(&key, n * 4)); CHECK(hipMalloc(&rank, n * 4));
    CHECK(hipMemcpy(px, x.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(py, y.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pz, z.data(), n * 4, hipMemcpyHostToDevice));
    printf("one particle in %d sits in a neighbouring tile\n", stray_one_in);
    for (int blocks : {2048}) {
        printf("%d workgroups, one contiguous run each:\n", blocks);
        printf("  loads + atomics + stores  %.3f ms\n", run<7, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  loads + stores            %.3f ms\n", run<5, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  loads + atomics           %.3f ms\n", run<3, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  atomics + stores          %.3f ms\n", run<6, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  loads only                %.3f ms\n", run<1, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  stores only               %.3f ms\n", run<4, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
        printf("  atomics only              %.3f ms\n", run<2, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, blocks));
    }
    printf("2048 workgroups, one contiguous run each, one atomic per distinct key of a wave:\n");
    printf("  loads + atomics + stores  %.3f ms\n", run<15, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, 2048));
    printf("  loads + stores            %.3f ms\n", run<13, false>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, 2048));
    printf("2048 workgroups, grid-stride:\n");
    printf("  loads + atomics + stores  %.3f ms\n", run<7, true>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, 2048));
    printf("  loads + stores            %.3f ms\n", run<5, true>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, 2048));
    printf("  atomics only              %.3f ms\n", run<2, true>(px, py, pz, n, per_tile, tc, key, rank, n_tiles, 2048));
    return 0;
}
