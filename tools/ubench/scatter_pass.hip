// Microbenchmark: the scatter pass of the re-bin.  24 arrays of n words move to the slots a stable sort
// by tile gives them; the store is sorted up to one particle in N that sits in a neighbouring tile
// (N = 4 when a real re-bin is due).  Compared: the identity permutation (a plain 24-array copy), the
// realistic one, and the realistic one with the writes made in destination order (gather form).
// Build: hipcc -O3 --offload-arch=gfx950 -o scatter_pass scatter_pass.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kSlots = 24;
struct Arrays { float *a[kSlots]; };

// as bin_scatter_kernel: thread j reads slot j of every array and writes slot d[j]
__global__ void __launch_bounds__(256) scatter_kernel(Arrays src, Arrays dst, const int *__restrict__ d, int n)
{
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)gridDim.x * 256) {
        const int t = d[j];
#pragma unroll
        for (int w = 0; w < kSlots; w++) dst.a[w][t] = src.a[w][j];
    }
}

// four consecutive source slots per thread: 16-byte loads, four 4-byte stores per array
__global__ void __launch_bounds__(256) scatter4_kernel(Arrays src, Arrays dst, const int *__restrict__ d, int n)
{
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n / 4; q += (long long)gridDim.x * 256) {
        const int4 t = reinterpret_cast<const int4 *>(d)[q];
#pragma unroll
        for (int w = 0; w < kSlots; w++) {
            const float4 v = reinterpret_cast<const float4 *>(src.a[w])[q];
            dst.a[w][t.x] = v.x; dst.a[w][t.y] = v.y; dst.a[w][t.z] = v.z; dst.a[w][t.w] = v.w;
        }
    }
}

// gather form: thread t writes slot t of every array from slot s[t]
__global__ void __launch_bounds__(256) gather_kernel(Arrays src, Arrays dst, const int *__restrict__ s, int n)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int j = s[t];
#pragma unroll
        for (int w = 0; w < kSlots; w++) dst.a[w][t] = src.a[w][j];
    }
}

template <typename K>
static float timed(K launch)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char **argv)
{
    const int n = 1 << 24, per_tile = 1900, n_tiles = 32 * 32 * 32;
    const int stray_one_in = argc > 1 ? atoi(argv[1]) : 4;
    std::vector<int> key(n), order(n), d(n), ident(n);
    unsigned s = 12345u;
    for (int j = 0; j < n; j++) {
        int t = j / per_tile;
        s = s * 1664525u + 1013904223u;
        if ((s >> 8) % stray_one_in == 0) {
            const int hop[6] = {1, n_tiles - 1, 32, n_tiles - 32, 1024, n_tiles - 1024};
            t = (t + hop[(s >> 20) % 6]) % n_tiles;
        }
        key[j] = t;
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int t = 0; t < n; t++) d[order[t]] = t;          // order[t] = source of destination t
    std::iota(ident.begin(), ident.end(), 0);
    Arrays src, dst;
    for (int w = 0; w < kSlots; w++) {
        CHECK(hipMalloc(&src.a[w], (size_t)n * 4)); CHECK(hipMalloc(&dst.a[w], (size_t)n * 4));
        CHECK(hipMemset(src.a[w], w, (size_t)n * 4));
    }
    int *dd, *ds, *di;
    CHECK(hipMalloc(&dd, (size_t)n * 4)); CHECK(hipMalloc(&ds, (size_t)n * 4)); CHECK(hipMalloc(&di, (size_t)n * 4));
    CHECK(hipMemcpy(dd, d.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(di, ident.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    printf("one particle in %d sits in a neighbouring tile; 24 x 2^24 words = 3.2 GB moved\n", stray_one_in);
    for (int blocks : {1024, 2048, 8192}) {
        printf("%d workgroups:\n", blocks);
        printf("  identity (plain copy)        %.3f ms\n", timed([&] { hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, di, n); }));
        printf("  scatter (reads in order)     %.3f ms\n", timed([&] { hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, dd, n); }));
        printf("  scatter, 4 slots per thread   %.3f ms\n", timed([&] { hipLaunchKernelGGL(scatter4_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, dd, n); }));
        printf("  gather (writes in order)     %.3f ms\n", timed([&] { hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, ds, n); }));
    }
    return 0;
}
