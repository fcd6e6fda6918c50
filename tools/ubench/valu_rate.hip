// Microbenchmark: issue rate of independent non-fused f32 multiply/add (and packed forms) per SIMD
// as a function of waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0: scalar mul/add mix, 1: v_pk_mul/v_pk_add via float2, 2: v_fma chain
__global__ void __launch_bounds__(256) valu_kernel(float *out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(a)); }
#pragma unroll
            for (int k = 0; k < 16; k++) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(b)); }
        } else if (MODE == 1) {
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v av = {a, a}, bv = {b, b};
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                float2v v = {x[k], x[k + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(av));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(bv));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(av));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(bv));
                x[k] = v.x; x[k + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(x[k]), "v"(a), "v"(b)); }
#pragma unroll
            for (int k = 0; k < 16; k++) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(x[k]), "v"(a), "v"(b)); }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += x[k];
    if (s == 123.456f) out[0] = s;
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    const char *names[] = {"v_mul_f32/v_add_f32", "v_pk_mul_f32/v_pk_add_f32", "v_fma_f32"};
    for (int mode = 0; mode < 3; mode++)
        for (int bpc : {1, 2, 3, 4, 8}) {           // 256-thread blocks per CU = waves per SIMD
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(valu_kernel<0>, dim3(256 * bpc), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                if (mode == 1) hipLaunchKernelGGL(valu_kernel<1>, dim3(256 * bpc), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                if (mode == 2) hipLaunchKernelGGL(valu_kernel<2>, dim3(256 * bpc), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double instr_per_simd = (double)bpc * iters * 32;      // wave-instructions per SIMD (pk counted once)
            if (mode == 1) instr_per_simd = (double)bpc * iters * 32;
            printf("%-28s waves/SIMD %d: %7.3f ms -> %5.2f ns per wave-instr per SIMD (= %4.2f cycles @2.4GHz), lane-ops/s %.1f T\n",
                   names[mode], bpc, best, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4,
                   instr_per_simd * 1024 * 64 * (mode == 1 ? 2 : 1) / (best * 1e-3) / 1e12);
        }
    return 0;
}
