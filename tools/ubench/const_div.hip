// Exhaustive check of a 3-instruction replacement for IEEE division by a constant:
//   q0 = x * rc;  r = fma(-c, q0, x);  q = fma(r, rc, q0)      with rc = RN(1 / c)
// against x / c for every one of the 2^32 binary32 bit patterns of x, for the divisors the
// step uses: the four values of `z` in triNoise3Dvec (src/common/noise.js:58, 73) and the
// fixed-point multiplier 1e7 (SIM:43, 282-284).  Prints the mismatch count per divisor, over all
// inputs and over the inputs the step can produce: tri() returns 0 or a value in [2^-25, 0.5]
// (|fract(x) - 0.5| with fract(x) a binary32 in [0, 1)), and decodeFixedPoint converts an i32,
// i.e. 0 or a magnitude in [1, 2^31].
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o const_div tools/ubench/const_div.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void check(float c, float rc, float lo, float hi, unsigned long long *bad, unsigned *first_bad)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (; i < (1ULL << 32); i += stride) {
        float x = __uint_as_float((unsigned)i);
        if (!(x == 0.0f || (fabsf(x) >= lo && fabsf(x) <= hi) || x != x)) continue;
        float want = x / c;
        float q0 = x * rc;
        float r = __builtin_fmaf(-c, q0, x);
        float got = __builtin_fmaf(r, rc, q0);
        bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        if (!same) {
            local++;
            atomicMin(first_bad, (unsigned)i);
        }
    }
    if (local) atomicAdd(bad, local);
}

int main()
{
    float z = 1.4f, divisors[5];
    for (int k = 0; k < 4; k++) { z = z * 1.5f; divisors[k] = z; }
    divisors[4] = 1.0e7f;
    unsigned long long *bad;
    unsigned *first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 4);
    int rc_all = 0;
    for (int pass = 0; pass < 10; pass++) {
        const int k = pass % 5;
        const bool domain = pass >= 5;
        const float lo = domain ? (k < 4 ? 0x1p-25f : 1.0f) : 0.0f;
        const float hi = domain ? (k < 4 ? 0.5f : 0x1p31f) : __builtin_inff();
        float c = divisors[k], rc = 1.0f / c;
        unsigned long long h = 0;
        unsigned f = 0xFFFFFFFFu;
        hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice);
        hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, c, rc, lo, hi, bad, first);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        float fx;
        memcpy(&fx, &f, 4);
        printf("divisor %.9g (rc %.9g), |x| in {0} U [%g, %g]: %llu mismatches", c, rc, lo, hi, h);
        if (h) printf(", smallest bit pattern 0x%08x (%g)", f, fx);
        printf("\n");
        rc_all |= h != 0;
    }
    return 0;
}
