// Device-side arithmetic shared by every kernel of the MLS-MPM step.
//
// Everything here is written so that the result is a pure function of IEEE-754
// binary32 multiply / add / subtract / divide / sqrt in a fixed order, with no
// fused multiply-add (the translation unit is built with -ffp-contract=off) and
// no reduced-precision intrinsic.  The grid is accumulated as wrapping i32 in the
// reference's fixed-point encoding, so any regrouping of the *integer* adds
// (LDS tiles, several blocks per tile, several GPUs) leaves the grid bit-identical.
//
// Reference lines are relative to the reference tree; SIM abbreviates
// src/mls-mpm/mlsMpmSimulator.js.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace mpm {

// Component slots of the SoA particle store (one float array of `capacity`
// elements per slot; slot kId is an int32 array).
enum : int {
    kPosX = 0, kPosY = 1, kPosZ = 2,
    kVelX = 3, kVelY = 4, kVelZ = 5,
    kC00 = 6,          // C[col][row] at kC00 + 3*col + row
    kDensity = 15, kMass = 16,
    kDirX = 17, kDirY = 18, kDirZ = 19,
    kColR = 20, kColG = 21, kColB = 22,
    kId = 23,
    kNumSlots = 24
};

// Uniforms as the kernels see them.  Grid sizes are the GLOBAL grid; the local
// grid of a slab starts at global plane x_off and holds local_nx planes.
struct Params {
    int gx, gy, gz;            // uniforms.gridSize (SIM:148)
    int x_off;                 // global x of local plane 0
    int local_nx;              // planes held locally
    float dt, stiffness, rest_density, dynamic_viscosity, noise, time;
    float ray_dir[3], ray_origin[3], mouse_force[3];
    unsigned cursor_interaction;
};

struct Vec3 { float x, y, z; };

// WGSL i32(f32): truncation toward zero, saturating, NaN -> 0.  v_cvt_i32_f32 has
// exactly these semantics.
__device__ __forceinline__ int f32_to_i32(float f)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// x / c for a compile-time constant c in three instructions instead of the dozen of an IEEE
// division: q0 = x * RN(1/c), one residual, one correction.  Bit-identical to x / c for
// x = +0 and every binary32 x with 2^-25 <= |x| <= 2^31, for the five divisors used here
// (checked over all 2^32 bit patterns by tools/ubench/const_div.hip; outside that range
// results that are subnormal can differ, and -0 yields +0).  Callers only pass values from
// that range: tri() returns 0 or [2^-25, 0.5], and decode_fixed converts an i32.
__device__ __forceinline__ float div_by_const(float x, float c)
{
    const float rc = 1.0f / c;
    const float q0 = x * rc;
    const float r = __builtin_fmaf(-c, q0, x);
    return __builtin_fmaf(r, rc, q0);
}

// SIM:279-281 encodeFixedPoint, fixedPointMultiplier = 1e7 (SIM:43).
__device__ __forceinline__ int encode_fixed(float f) { return f32_to_i32(f * 1.0e7f); }
// SIM:282-284 decodeFixedPoint.
__device__ __forceinline__ float decode_fixed(int i) { return div_by_const((float)i, 1.0e7f); }

__device__ __forceinline__ float fract(float x) { return x - floorf(x); }

// Quadratic B-spline weights (SIM:307-312, 347-352, 1322-1328).
struct Stencil {
    int cx, cy, cz;            // cellIndex = ivec3(pos) - 1 (global cell coordinates)
    float wx[3], wy[3], wz[3];
    float dx[3], dy[3], dz[3]; // cellDist per axis: (float(cell) + 0.5) - pos
};

__device__ __forceinline__ void axis_weights(float p, int &ci, float w[3], float d[3])
{
    ci = f32_to_i32(p) - 1;
    float cd = fract(p) - 0.5f;
    w[0] = (0.5f * (0.5f - cd)) * (0.5f - cd);
    w[1] = 0.75f - cd * cd;
    w[2] = (0.5f * (0.5f + cd)) * (0.5f + cd);
#pragma unroll
    for (int g = 0; g < 3; g++) d[g] = ((float)(ci + g) + 0.5f) - p;
}

__device__ __forceinline__ Stencil make_stencil(float px, float py, float pz)
{
    Stencil s;
    axis_weights(px, s.cx, s.wx, s.dx);
    axis_weights(py, s.cy, s.wy, s.dy);
    axis_weights(pz, s.cz, s.wz, s.dz);
    return s;
}

// src/common/noise.js:3-13.  NONNEG: the caller guarantees x is +0 or a positive normal number;
// v_fract_f32 then returns exactly x - floor(x) (checked over every such bit pattern by
// tools/ubench/fract_check.hip; the instruction clamps below 1, which only differs for tiny
// negative x) and saves the separate floor.
template <bool NONNEG>
__device__ __forceinline__ float tri_t(float x)
{
    return fabsf((NONNEG ? __builtin_amdgcn_fractf(x) : fract(x)) - 0.5f);
}
__device__ __forceinline__ float tri(float x) { return tri_t<false>(x); }

// src/common/noise.js:49-80.  Parameter names follow the reference signature; the
// g2p call site (SIM:1318) passes the clock as `speed` and 0.11 as `time`.
template <bool NONNEG>
__device__ __forceinline__ Vec3 tri_noise3d_t(Vec3 position, float s)
{
    float px = position.x, py = position.y, pz = position.z;
    float bx = px, by = py, bz = pz;
    float z = 1.4f;
    float rx = 0.0f, ry = 0.0f, rz = 0.0f;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        float b2x = bx * 2.0f, b2y = by * 2.0f, b2z = bz * 2.0f;
        // tri3 (noise.js:27-37)
        float dgx = tri_t<NONNEG>(b2z + tri_t<NONNEG>(b2y * 1.0f));
        float dgy = tri_t<NONNEG>(b2z + tri_t<NONNEG>(b2x * 1.0f));
        float dgz = tri_t<NONNEG>(b2y + tri_t<NONNEG>(b2x * 1.0f));
        px = px + (dgx + s); py = py + (dgy + s); pz = pz + (dgz + s);
        bx = bx * 1.8f; by = by * 1.8f; bz = bz * 1.8f;
        z = z * 1.5f;
        px = px * 1.2f; py = py * 1.2f; pz = pz * 1.2f;
        float i0 = tri_t<NONNEG>(py), i1 = tri_t<NONNEG>(pz), i2 = tri_t<NONNEG>(px);
        float m0 = tri_t<NONNEG>(px + i0), m1 = tri_t<NONNEG>(py + i1), m2 = tri_t<NONNEG>(pz + i2);
        float t0 = tri_t<NONNEG>(pz + m0), t1 = tri_t<NONNEG>(px + m1), t2 = tri_t<NONNEG>(py + m2);
        // z is a compile-time constant here (the loop is unrolled): 2.1, 3.15, 4.725, 7.0875
        rx = rx + div_by_const(t0, z); ry = ry + div_by_const(t1, z); rz = rz + div_by_const(t2, z);
        bx = bx + 0.14f; by = by + 0.14f; bz = bz + 0.14f;
    }
    return Vec3{rx, ry, rz};
}

__device__ __forceinline__ Vec3 tri_noise3d(Vec3 position, float speed, float time)
{
    const float s = time * (0.1f * speed);
    // Every argument of tri() is a sum or product of the position components, s, constants and
    // earlier tri() values (all >= 0), so when those four are +0 or positive normal numbers --
    // any particle inside the grid, any non-negative clock -- so is every argument.
    constexpr int kPosZeroOrNormal = (1 << 6) | (1 << 8);
    const bool nonneg = __builtin_amdgcn_class(position.x, kPosZeroOrNormal) &&
                        __builtin_amdgcn_class(position.y, kPosZeroOrNormal) &&
                        __builtin_amdgcn_class(position.z, kPosZeroOrNormal) &&
                        __builtin_amdgcn_class(s, kPosZeroOrNormal);
    return nonneg ? tri_noise3d_t<true>(position, s) : tri_noise3d_t<false>(position, s);
}

__device__ __forceinline__ Vec3 normalize3(Vec3 v)
{
    float len = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    return Vec3{v.x / len, v.y / len, v.z / len};
}

// Per-particle state p2g2 derives before its scatter loop (SIM:365-375):
// updates the stored density and returns eq16Term0 as 6 unique entries of the
// symmetric 3x3 matrix (C + C^T is bitwise symmetric, so stress is too).
struct Eq16 { float m00, m11, m22, m01, m02, m12; };

__device__ __forceinline__ Eq16 p2g2_stress(float density, const float C[9], const Params &P)
{
    float volume = 1.0f / density;
    float ratio = density / P.rest_density;
    float pow5 = ((ratio * ratio) * (ratio * ratio)) * ratio;
    float pressure = fmaxf(0.0f, (pow5 - 1.0f) * P.stiffness);
    float k = volume * -4.0f;
    float visc = P.dynamic_viscosity;
    // C[3*col + row]; strain[c][r] = C[c][r] + C[r][c]
    float s00 = -pressure + (C[0] + C[0]) * visc;
    float s11 = -pressure + (C[4] + C[4]) * visc;
    float s22 = -pressure + (C[8] + C[8]) * visc;
    float s01 = 0.0f + (C[3] + C[1]) * visc;   // col 1,row 0  == col 0,row 1
    float s02 = 0.0f + (C[6] + C[2]) * visc;
    float s12 = 0.0f + (C[7] + C[5]) * visc;
    Eq16 e;
    e.m00 = (k * s00) * P.dt; e.m11 = (k * s11) * P.dt; e.m22 = (k * s22) * P.dt;
    e.m01 = (k * s01) * P.dt; e.m02 = (k * s02) * P.dt; e.m12 = (k * s12) * P.dt;
    return e;
}

// Everything g2p does to a particle apart from the 27-cell gather
// (SIM:484-494, 1316-1320, 1354-1362, 1515-1533).
struct G2PIn {
    float px, py, pz;
    float mass;
    float dirx, diry, dirz;
};
struct G2POut {
    float px, py, pz, vx, vy, vz;
    float C[9];
    float dirx, diry, dirz;
};

// Forces applied before the gather: centre gravity (SIM:490-492) and noise
// (SIM:1317-1320).  Returns the initial particleVelocity.
__device__ __forceinline__ Vec3 g2p_pre_forces(float px, float py, float pz, const Params &P)
{
    Vec3 v{0.0f, 0.0f, 0.0f};
    Vec3 pn = normalize3(Vec3{px / (float)(P.gx - 1) - 0.5f,
                              py / (float)(P.gy - 1) - 0.5f,
                              pz / (float)(P.gz - 1) - 0.5f});
    v.x = v.x - (pn.x * 0.1f) * P.dt;
    v.y = v.y - (pn.y * 0.1f) * P.dt;
    v.z = v.z - (pn.z * 0.1f) * P.dt;
    Vec3 n = tri_noise3d(Vec3{px * 0.015f, py * 0.015f, pz * 0.015f}, P.time, 0.11f);
    n = normalize3(Vec3{n.x - 0.285f, n.y - 0.285f, n.z - 0.285f});
    n = Vec3{n.x * 0.28f, n.y * 0.28f, n.z * 0.28f};
    v.x = v.x - (n.x * P.noise) * P.dt;
    v.y = v.y - (n.y * P.noise) * P.dt;
    v.z = v.z - (n.z * P.noise) * P.dt;
    return v;
}

__device__ __forceinline__ float wall(float xn, float v, float wall_max)
{
    if (xn < 3.0f) v = v + (3.0f - xn) * 0.3f;
    if (xn > wall_max) v = v + (wall_max - xn) * 0.3f;
    return v;
}

// The mouse force (SIM:1354-1360).
__device__ __forceinline__ Vec3 g2p_mouse(const G2PIn &in, Vec3 v, const Params &P)
{
    if (P.cursor_interaction == 1u) {
        float bx = in.px * 1.0f - P.ray_origin[0];
        float by = in.py * 1.0f - P.ray_origin[1];
        float bz = in.pz * 0.4f - P.ray_origin[2];
        float cx = P.ray_dir[1] * bz - P.ray_dir[2] * by;
        float cy = P.ray_dir[2] * bx - P.ray_dir[0] * bz;
        float cz = P.ray_dir[0] * by - P.ray_dir[1] * bx;
        float dist = sqrtf((cx * cx + cy * cy) + cz * cz);
        float f = fmaxf(1.0f - dist * 0.1f, 0.0f);
        float force = f * f;
        v.x = v.x + (P.mouse_force[0] * 1.0f) * force;
        v.y = v.y + (P.mouse_force[1] * 1.0f) * force;
        v.z = v.z + (P.mouse_force[2] * 1.0f) * force;
        v.x = v.x * in.mass; v.y = v.y * in.mass; v.z = v.z * in.mass;
    }
    return v;
}

// C = B * 4, integration, clamp, soft walls, direction (SIM:1362, 1515-1533).
// B[3*col + row] is the gathered affine sum.
__device__ __forceinline__ G2POut g2p_integrate(const G2PIn &in, Vec3 v, const float B[9], const Params &P)
{
    float px = in.px, py = in.py, pz = in.pz;
    G2POut o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.C[k] = B[k] * 4.0f;          // SIM:1362
    px = px + v.x * P.dt; py = py + v.y * P.dt; pz = pz + v.z * P.dt;   // SIM:1515
    px = fminf(fmaxf(px, 2.0f), (float)(P.gx - 2));             // SIM:1516
    py = fminf(fmaxf(py, 2.0f), (float)(P.gy - 2));
    pz = fminf(fmaxf(pz, 2.0f), (float)(P.gz - 2));
    float xnx = px + (v.x * P.dt) * 3.0f;                        // SIM:1519
    float xny = py + (v.y * P.dt) * 3.0f;
    float xnz = pz + (v.z * P.dt) * 3.0f;
    v.x = wall(xnx, v.x, (float)P.gx - 3.0f);                    // SIM:1520-1527
    v.y = wall(xny, v.y, (float)P.gy - 3.0f);
    v.z = wall(xnz, v.z, (float)P.gz - 3.0f);
    o.px = px; o.py = py; o.pz = pz;
    o.vx = v.x; o.vy = v.y; o.vz = v.z;
    o.dirx = in.dirx * (1.0f - 0.1f) + v.x * 0.1f;               // SIM:1532-1533
    o.diry = in.diry * (1.0f - 0.1f) + v.y * 0.1f;
    o.dirz = in.dirz * (1.0f - 0.1f) + v.z * 0.1f;
    return o;
}

// Everything after the gather on the chaos path.
__device__ __forceinline__ G2POut g2p_finish(const G2PIn &in, Vec3 v, const float B[9], const Params &P)
{
    return g2p_integrate(in, g2p_mouse(in, v, P), B, P);
}

}  // namespace mpm
