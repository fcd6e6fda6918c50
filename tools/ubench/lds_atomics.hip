// Microbenchmark: LDS integer atomic / store throughput on gfx950 under different address
// patterns, and global i32 atomic flush rate.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kWords = 8192;   // 32 KB of LDS

template <int MODE>   // 0 ds_add_u32, 1 ds_write_b32, 2 ds_add_u64, 3 ds_add_rtn_u32
__global__ void __launch_bounds__(256) lds_kernel(const int *__restrict__ idx, int iters, int *out)
{
    __shared__ int lds[kWords];
    for (int i = threadIdx.x; i < kWords; i += 256) lds[i] = 0;
    __syncthreads();
    int a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = idx[k * 256 + threadIdx.x];
    int acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) atomicAdd(&lds[a[k]], it + k);
            if (MODE == 1) lds[a[k]] = it + k;
            if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (a[k] >> 1), (unsigned long long)(it + k));
            if (MODE == 3) acc += atomicAdd(&lds[a[k]], it + k);
        }
        if (MODE == 1) asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (out && (lds[threadIdx.x] + acc) == 0x7fffffff) out[0] = 1;
}

__global__ void __launch_bounds__(256) global_atomic_kernel(int *__restrict__ grid, long long words, int stride_words, int reps)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; r++)
        for (long long i = t; i < words; i += total) atomicAdd(&grid[(i * stride_words) % words], 1);
}

int main()
{
    const int blocks = 256 * 8, iters = 2000;
    std::vector<std::vector<int>> patterns;
    std::vector<const char *> names;
    auto add = [&](const char *n, std::vector<int> v) { names.push_back(n); patterns.push_back(v); };
    std::vector<int> v(8 * 256);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (k * 256 + t) % kWords;
    add("consecutive (conflict-free)", v);
    srand(1);
    for (auto &x : v) x = rand() % 4000;
    add("random in 4000 words", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = ((t / 4) * 4 + k * 64) % kWords;
    add("4 lanes per address", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (t * 2 + k) % kWords;
    add("stride 2 words", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (t * 32 + k) % kWords;
    add("stride 32 words (same bank)", v);
    // which lanes conflict: addresses with random rows and a prescribed residue
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (rand() % 120) * 32 + (t % 32);
    add("distinct mod 32 per 32 lanes", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (rand() % 60) * 64 + (t % 64);
    add("distinct mod 64 per 64 lanes", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (rand() % 120) * 32 + ((t % 16) * 2 + (t / 32) % 2);
    add("distinct mod 32 per 16 lanes", v);
    for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) v[k * 256 + t] = (rand() % 30) * 128 + (t % 64) * 2;
    add("distinct mod 128, even words", v);
    int *d_idx, *d_out;
    CHECK(hipMalloc(&d_idx, 8 * 256 * sizeof(int)));
    CHECK(hipMalloc(&d_out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *modes[] = {"ds_add_u32", "ds_write_b32", "ds_add_u64", "ds_add_rtn_u32"};
    for (size_t p = 0; p < patterns.size(); p++) {
        CHECK(hipMemcpy(d_idx, patterns[p].data(), 8 * 256 * sizeof(int), hipMemcpyHostToDevice));
        for (int m = 0; m < 4; m++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (m == 0) hipLaunchKernelGGL(lds_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_out);
                if (m == 1) hipLaunchKernelGGL(lds_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_out);
                if (m == 2) hipLaunchKernelGGL(lds_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_out);
                if (m == 3) hipLaunchKernelGGL(lds_kernel<3>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double wave_instr = (double)blocks * 4 * iters * 8;
            // cycles per wave-instruction per CU at 2.4 GHz, 256 CUs
            printf("%-30s %-16s %8.3f ms  %6.2f cycles/wave-instr/CU  %7.2f G lane-ops/s\n", names[p], modes[m], best,
                   best * 1e-3 * 2.4e9 * 256 / wave_instr, wave_instr * 64 / (best * 1e-3) / 1e9);
        }
    }
    // global i32 atomics: contiguous (each wave-instruction = 256 contiguous bytes) and 16-byte strided
    long long words = 64LL << 20;   // 256 MB
    int *grid;
    CHECK(hipMalloc(&grid, words * 4));
    CHECK(hipMemset(grid, 0, words * 4));
    for (int stride : {1, 4, 17, 1031}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(global_atomic_kernel, dim3(256 * 16), dim3(256), 0, 0, grid, words, stride, 1);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("global atomicAdd i32, lane stride %4d words: %8.3f ms  %7.1f GB/s of added bytes  %6.2f G atomics/s\n", stride,
               best, words * 4 / (best * 1e-3) / 1e9, words / (best * 1e-3) / 1e9);
    }
    return 0;
}
