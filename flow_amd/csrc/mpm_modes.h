// The mode branches of g2p (SIM:444-1313 forces, SIM:1365-1476 dampings): image-grid
// attraction, front gravity, cylinder, sphere and polygon containment, colour spheres and
// the nine population-segment modes.  The g2p kernels are instantiated twice; the plain
// instantiation never touches anything in this file, so the chaos path keeps its register
// budget.  Same arithmetic rules as mpm_device.h: IEEE binary32, fixed order, no FMA.
#pragma once

#include "../../include/flowmpm.h"
#include "mpm_device.h"

namespace mpm {

// What the mode instantiation reads from device memory (one record per handle).
struct ModeParams {
    flowmpm_mode_uniforms u;
    unsigned grid_mode;            // uniforms.gridMode (SIM:152)
    unsigned front_gravity_mode;   // uniforms.frontGravityMode (SIM:153)
    float hide_percentage;         // uniforms.hidePercentage (SIM:260)
    unsigned num_particles;        // uniforms.numParticles (SIM:151)
};

// SIM:441-442: the first hidePercentage percent of the buffer are "hidden": no mode force, no
// gather, no mouse force, parked at the x faces instead.
__device__ __forceinline__ bool mode_hidden(int index, const ModeParams *__restrict__ M)
{
    const float index_percent = ((float)(unsigned)index / (float)M->num_particles) * 100.0f;
    return M->hide_percentage > 0.0f && index_percent < M->hide_percentage;
}

// Colours the population-segment modes compare particle colours with (SIM:766, 847, 906,
// 965, 1024, 1083, 1142, 1201, 1260), in the order of flowmpm_mode_uniforms.
__device__ const float kSegmentColor[FLOWMPM_SEGMENTS][3] = {
    {1.0f, 0.3372549f, 0.717647f}, {0.737f, 0.588f, 0.976f}, {0.773f, 0.941f, 0.949f},
    {0.0f, 0.455f, 0.529f},        {0.082f, 0.224f, 0.263f}, {0.137f, 0.137f, 0.337f},
    {0.314f, 0.071f, 0.102f},      {1.0f, 0.392f, 0.145f},   {0.839f, 0.957f, 0.600f},
};

__device__ __forceinline__ float length3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// clamp(boostTime / boostDuration, 0, 1) through smoothstep (SIM:751-753 and its copies).
__device__ __forceinline__ float eased_boost(float boost_time, float boost_duration)
{
    float bp = clampf(boost_time / boost_duration, 0.0f, 1.0f);
    return (bp * bp) * (3.0f - bp * 2.0f);
}

// v += (direction * strength) * dt
__device__ __forceinline__ void mode_push(Vec3 &v, float dx, float dy, float dz, float strength, float dt)
{
    v.x = v.x + (dx * strength) * dt;
    v.y = v.y + (dy * strength) * dt;
    v.z = v.z + (dz * strength) * dt;
}

// Outside the sphere: pushed back in (SIM:517-528, 535-546, 740-758, 773-794, ...).
// boosted: strength = (excess * 0.3) * mix(1, 20, eased); plain: strength = excess * 0.3.
__device__ __forceinline__ void contain_in_sphere(Vec3 &v, float px, float py, float pz, const float *center,
                                                  float radius, bool boosted, float eased, float dt)
{
    float tx = px - center[0], ty = py - center[1], tz = pz - center[2];
    float dist = length3(tx, ty, tz);
    if (dist > radius) {
        float excess = dist - radius;
        float strength = excess * 0.3f;
        if (boosted) strength = strength * (1.0f * (1.0f - eased) + 20.0f * eased);
        mode_push(v, -(tx / dist), -(ty / dist), -(tz / dist), strength, dt);
    }
}

// Inside the negative-space sphere: pushed out (SIM:821-840 and its copies).
__device__ __forceinline__ void repel_from_sphere(Vec3 &v, float px, float py, float pz, const float *center,
                                                  float radius, float eased, float dt)
{
    float tx = px - center[0], ty = py - center[1], tz = pz - center[2];
    float dist = length3(tx, ty, tz);
    if (dist < radius) {
        float depth = radius - dist;
        float strength = (depth * 0.3f) * (1.0f * (1.0f - eased) + 20.0f * eased);
        mode_push(v, tx / dist, ty / dist, tz / dist, strength, dt);
    }
}

// Everything g2p adds to particleVelocity before the gather when a mode is on
// (SIM:444-1320).  `index` is the reference's instanceIndex: the particle's slot in the
// host's buffer, which the store carries in its id slot.
__device__ __forceinline__ Vec3 g2p_pre_forces_modes(float px, float py, float pz, float cr, float cg, float cb,
                                                     int index, bool hide, const Params &P,
                                                     const ModeParams *__restrict__ M)
{
    const flowmpm_mode_uniforms &m = M->u;
    const float dt = P.dt;
    Vec3 v{0.0f, 0.0f, 0.0f};

    if (M->grid_mode == 1u && !hide) {              // SIM:445-482
        const int gw = (int)m.grid_width;
        const int gxi = gw ? index % gw : 0;        // WGSL: x % 0 = 0, x / 0 = x
        const int gyi = gw ? index / gw : index;
        float hash = ((float)index * 12.9898f + (float)gxi * 78.233f) + (float)gyi * 37.719f;
        float random_x = fract(hash) * 2.0f - 1.0f;
        float random_y = fract(hash * 43758.5453f) * 2.0f - 1.0f;
        float offset_x = (random_x * m.grid_spacing_x) * m.grid_random_offset;
        float offset_y = (random_y * m.grid_spacing_y) * m.grid_random_offset;
        float tx = (m.grid_start_x + (float)gxi * m.grid_spacing_x) + offset_x;
        float ty = (m.grid_start_y + (float)gyi * m.grid_spacing_y) + offset_y;
        float tz = m.grid_z;
        Vec3 n = tri_noise3d(Vec3{px * 0.92f, py * 0.92f, pz * 0.92f}, 1.15f, P.time);
        tx = tx + (n.x * m.grid_noise_strength) * 2.0f;
        ty = ty + (n.y * m.grid_noise_strength) * 2.0f;
        tz = tz + (n.z * m.grid_noise_strength) * 2.0f;
        Vec3 zn = tri_noise3d(Vec3{px * 0.85f + 100.0f, py * 0.85f + 200.0f, pz * 0.85f + 300.0f}, 0.02f, P.time);
        tz = tz + (zn.z * m.grid_noise_strength) * 2.0f;
        float ox = tx - px, oy = ty - py, oz = tz - pz;
        float dist = length3(ox, oy, oz);
        mode_push(v, ox / dist, oy / dist, oz / dist, dist * 10.0f, dt);
    } else if (M->front_gravity_mode == 1u && !hide) {   // SIM:485-488
        v.x = v.x + (0.0f * 0.1f) * dt;
        v.y = v.y + (0.0f * 0.1f) * dt;
        v.z = v.z + (-1.0f * 0.1f) * dt;
    } else {                                        // SIM:490-492
        Vec3 pn = normalize3(Vec3{px / (float)(P.gx - 1) - 0.5f, py / (float)(P.gy - 1) - 0.5f,
                                  pz / (float)(P.gz - 1) - 0.5f});
        v.x = v.x - (pn.x * 0.1f) * dt;
        v.y = v.y - (pn.y * 0.1f) * dt;
        v.z = v.z - (pn.z * 0.1f) * dt;
    }

    if (m.cylinder_mode == 1u && !hide) {           // SIM:497-511
        float tx = px - m.cylinder_center_x, ty = py - m.cylinder_center_y;
        float dist = length3(tx, ty, 0.0f);
        if (dist < m.cylinder_radius) {
            float strength = (1.0f - dist / m.cylinder_radius) * 0.5f;
            mode_push(v, tx / dist, ty / dist, 0.0f / dist, strength, dt);
        }
    }
    if (m.sphere_containment_mode == 1u && !hide)   // SIM:514-529
        contain_in_sphere(v, px, py, pz, m.sphere_center, m.sphere_radius, false, 0.0f, dt);
    if (m.two_color_sphere_mode == 1u && !hide)     // SIM:532-547
        contain_in_sphere(v, px, py, pz, m.sphere_center, m.sphere_radius, false, 0.0f, dt);

    if (m.polygon_containment_mode == 1u && !hide) {    // SIM:550-696
        const int count = (int)m.polygon_vertex_count;
        int inside = 0;
        for (int i = 0; i < count; i++) {
            const int j = (i == count - 1) ? 0 : i + 1;
            const float *v0 = m.polygon_vertices[i < 15 ? i : 15];
            const float *v1 = m.polygon_vertices[j < 15 ? j : 15];
            const float v0x = v0[0], v0y = v0[1], v1x = v1[0], v1y = v1[1];
            if ((v0y > py) != (v1y > py)) {
                float t = (py - v0y) / (v1y - v0y);
                float x_intersect = v0x + (v1x - v0x) * t;
                if (x_intersect > px) inside = (inside + 1) % 2;
            }
        }
        if (inside == 0) {
            float min_dist = 9999.0f, nx = 0.0f, ny = 0.0f;
            for (int i = 0; i < count; i++) {
                const int j = (i == count - 1) ? 0 : i + 1;
                const float *v0 = m.polygon_vertices[i < 15 ? i : 15];
                const float *v1 = m.polygon_vertices[j < 15 ? j : 15];
                const float v0x = v0[0], v0y = v0[1];
                float ex = v1[0] - v0x, ey = v1[1] - v0y;
                float tpx = px - v0x, tpy = py - v0y;
                float edge_len = sqrtf(ex * ex + ey * ey);
                float edge_len_sq = edge_len * edge_len;
                float t = (edge_len_sq < 0.0001f) ? 0.0f : clampf((tpx * ex + tpy * ey) / edge_len_sq, 0.0f, 1.0f);
                float qx = v0x + ex * t, qy = v0y + ey * t;
                float ddx = px - qx, ddy = py - qy;
                float dist = sqrtf(ddx * ddx + ddy * ddy);
                if (dist < min_dist) { min_dist = dist; nx = qx; ny = qy; }
            }
            float ox = nx - px, oy = ny - py, oz = pz - pz;
            float dist = length3(ox, oy, oz);
            mode_push(v, ox / dist, oy / dist, oz / dist, dist * 0.3f, dt);
        }
    }

    if (m.color_spheres_mode == 1u && !hide) {      // SIM:699-759
        float min_dist = 999.0f;
        int closest = 0;
        for (int i = 0; i < 10; i++) {
            const float *pc = m.palette_colors[i < 9 ? i : 9];
            float dist = length3(cr - pc[0], cg - pc[1], cb - pc[2]);
            if (dist < min_dist) { min_dist = dist; closest = i; }
        }
        contain_in_sphere(v, px, py, pz, m.color_sphere_positions[closest < 9 ? closest : 9], m.color_sphere_radius,
                          true, eased_boost(m.color_spheres_boost_time, m.color_spheres_boost_duration), dt);
    }

    for (int k = 0; k < FLOWMPM_SEGMENTS; k++) {    // SIM:762-1313
        if (m.segment_mode[k] != 1u || hide) continue;
        const bool is_segment =
            length3(cr - kSegmentColor[k][0], cg - kSegmentColor[k][1], cb - kSegmentColor[k][2]) < 0.1f;
        const float eased = eased_boost(m.segment_boost_time[k], m.segment_boost_duration[k]);
        contain_in_sphere(v, px, py, pz, m.sphere_center, m.segment_main_radius[k], true, eased, dt);
        if (is_segment)
            contain_in_sphere(v, px, py, pz, m.segment_sphere_center[k], m.segment_sphere_radius[k], true, eased, dt);
        repel_from_sphere(v, px, py, pz, m.segment_negative_center[k], m.negative_space_sphere_radius, eased, dt);
    }

    if (M->grid_mode == 0u && M->front_gravity_mode == 0u) {   // SIM:1316-1320
        Vec3 n = tri_noise3d(Vec3{px * 0.015f, py * 0.015f, pz * 0.015f}, P.time, 0.11f);
        n = normalize3(Vec3{n.x - 0.285f, n.y - 0.285f, n.z - 0.285f});
        n = Vec3{n.x * 0.28f, n.y * 0.28f, n.z * 0.28f};
        v.x = v.x - (n.x * P.noise) * dt;
        v.y = v.y - (n.y * P.noise) * dt;
        v.z = v.z - (n.z * P.noise) * dt;
    }
    return v;
}

// The velocity dampings applied after C is stored (SIM:1365-1476).
__device__ __forceinline__ Vec3 mode_damping(Vec3 v, bool hide, const ModeParams *__restrict__ M)
{
    const flowmpm_mode_uniforms &m = M->u;
    if (M->front_gravity_mode == 1u) { v.x = v.x * 0.98f; v.y = v.y * 0.98f; v.z = v.z * 0.98f; }
    if (m.color_spheres_mode == 1u && !hide) {
        float e = eased_boost(m.color_spheres_boost_time, m.color_spheres_boost_duration);
        float damping = 0.98f * (1.0f - e) + 0.7f * e;
        v.x = v.x * damping; v.y = v.y * damping; v.z = v.z * damping;
    }
    for (int k = 0; k < FLOWMPM_SEGMENTS; k++) {
        if (m.segment_mode[k] != 1u || hide) continue;
        float e = eased_boost(m.segment_boost_time[k], m.segment_boost_duration[k]);
        float damping = 0.98f * (1.0f - e) + 0.7f * e;
        v.x = v.x * damping; v.y = v.y * damping; v.z = v.z * damping;
    }
    return v;
}

// Everything after the gather when a mode is on: the mouse force only outside grid mode and
// for visible particles (SIM:1355), the dampings, the parking of hidden particles, then the
// common integration.
//
// Parking (SIM:1479-1513): `gridSize.x.mul(0.1)` has an int left operand, and three.js types a
// binary operator by its left operand unless the right one is a longer vector, building both
// operands in that type.  The literal becomes i32(0.1) = 0, the "edge region" is 0 wide and the
// targets are the planes x = 0 and x = gridSize.x; the `.div(2.0)` on these ints divides by 2.
__device__ __forceinline__ G2POut g2p_finish_modes(const G2PIn &in, Vec3 v, const float B[9], int index, bool hide,
                                                   const Params &P, const ModeParams *__restrict__ M)
{
    if (M->grid_mode == 0u && !hide) v = g2p_mouse(in, v, P);
    v = mode_damping(v, hide, M);
    if (hide) {
        const int edge_region_width = P.gx * (int)0.1f;
        const int left_edge_max = edge_region_width, right_edge_min = P.gx - edge_region_width;
        const float hash = (float)(unsigned)index * 12.9898f;
        const bool use_left_edge = fract(hash) < 0.5f;
        const int target_x = use_left_edge ? left_edge_max / 2 : right_edge_min + edge_region_width / 2;
        const float dist = fabsf(in.px - (float)target_x);
        if (dist > (float)(edge_region_width / 2)) {
            const float toward = in.px < (float)target_x ? 1.0f : -1.0f;
            const float excess = dist - (float)(edge_region_width / 2);
            const float strength = excess * 0.3f;
            v.x = v.x + (toward * strength) * P.dt;
        }
    }
    return g2p_integrate(in, v, B, P);
}

}  // namespace mpm
