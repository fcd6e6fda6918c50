// Exhaustive check of v_fract_f32 against the WGSL definition fract(x) = x - floor(x)
// (src/common/noise.js:3-13 builds tri() on it), over every binary32 bit pattern.  Prints the
// mismatch count over all inputs and over the inputs tri() sees when positions and the clock
// are not negative: +0 and the positive normal numbers.  The instruction clamps its result
// below 1, which differs from x - floor(x) only for tiny negative x (where the subtraction
// rounds to 1.0).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o fract_check tools/ubench/fract_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void check(int domain, unsigned long long *bad, unsigned *first_bad)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (; i < (1ULL << 32); i += stride) {
        const unsigned u = (unsigned)i;
        float x = __uint_as_float(u);
        if (domain) {
            const unsigned e = (u >> 23) & 0xFFu;
            const bool ok = (u >> 31) == 0 && (u == 0 || (e >= 1 && e <= 254));   // +0 or positive normal
            if (!ok) continue;
        }
        float want = x - floorf(x);
        float got = __builtin_amdgcn_fractf(x);
        bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        if (!same) {
            local++;
            atomicMin(first_bad, u);
        }
    }
    if (local) atomicAdd(bad, local);
}

int main()
{
    unsigned long long *bad;
    unsigned *first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 4);
    for (int domain = 0; domain < 2; domain++) {
        unsigned long long h = 0;
        unsigned f = 0xFFFFFFFFu;
        hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice);
        hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, domain, bad, first);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        float fx;
        memcpy(&fx, &f, 4);
        printf("%s: %llu mismatches", domain ? "+0 and positive normal x" : "all x", h);
        if (h) printf(", smallest bit pattern 0x%08x (%g)", f, fx);
        printf("\n");
    }
    return 0;
}
