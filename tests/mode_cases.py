"""Mode-uniform test cases shared by the oracle tests, the GPU parity tests and the golden
vector generator.  A case is (uniform keyword overrides, mode-uniform field values); `fill`
writes the field values into either ctypes mirror (the oracle's or the library's: same names)."""
import numpy as np

from flow_amd import modes as host_modes
from flow_amd.particles import palette_rgb

GRID = (128, 64, 64)       # the mode geometry (sphere centre, polygon window) assumes this box

_BASE = dict(
    cylinder_radius=44.0, cylinder_center_x=64.0, cylinder_center_y=32.0,
    sphere_radius=25.0, sphere_center=(64.0, 32.0, 32.0), color_sphere_radius=7.0,
    color_spheres_boost_duration=1.4, negative_space_sphere_radius=17.0,
    segment_boost_duration=(1.4,) * 9, segment_main_radius=(18.0,) * 9, segment_sphere_radius=(7.0,) * 9,
    grid_z=32.0, grid_random_offset=0.5, grid_noise_strength=3.5,
)


def _segment_geometry():
    centers, negatives = [], []
    for sphere_deg, chunk_deg in host_modes.SEGMENT_ANGLES:
        a, b = np.deg2rad(sphere_deg), np.deg2rad(chunk_deg)
        centers.append((64 + 40 * np.cos(a), 32 + 40 * np.sin(a), 32.0))
        negatives.append((64 + 20 * np.cos(b), 32 + 20 * np.sin(b), 32.0))
    return centers, negatives


def _polygon():
    v = host_modes.polygon_vertices()
    return v + [(0.0, 0.0)] * (16 - len(v)), len(v)


def _color_spheres():
    positions, unique = host_modes.color_sphere_layout()
    pos = [positions[i] if i < len(positions) else (0.0, 0.0, 0.0) for i in range(13)]
    pal = [tuple(unique[min(i, len(unique) - 1)]) for i in range(13)]
    return pos, pal


def case(name):
    """-> (uniform overrides, mode fields)."""
    uni, m = {}, dict(_BASE)
    centers, negatives = _segment_geometry()
    m["segment_sphere_center"], m["segment_negative_center"] = centers, negatives
    if name == "front_gravity":
        uni["front_gravity_mode"] = 1
    elif name == "front_gravity_cylinder":
        uni["front_gravity_mode"] = 1
        m["cylinder_mode"] = 1
    elif name == "sphere":
        m["sphere_containment_mode"] = 1
    elif name == "two_color_sphere":
        m["two_color_sphere_mode"] = 1
    elif name == "polygon":
        m["polygon_containment_mode"] = 1
        m["polygon_vertices"], m["polygon_vertex_count"] = _polygon()
    elif name == "polygon_16":
        # a 16-gon, so that the capped vertex selects (index 15) are exercised
        ang = np.linspace(0, 2 * np.pi, 16, endpoint=False)
        m["polygon_containment_mode"] = 1
        m["polygon_vertices"] = [(64 + 30 * np.cos(a), 32 + 20 * np.sin(a)) for a in ang]
        m["polygon_vertex_count"] = 16
    elif name == "polygon_empty":
        m["polygon_containment_mode"] = 1
        m["polygon_vertex_count"] = 0
    elif name == "color_spheres":
        m["color_spheres_mode"] = 1
        m["color_spheres_boost_time"] = 0.9
        m["color_sphere_positions"], m["palette_colors"] = _color_spheres()
    elif name.startswith("segment_"):
        k = int(name.split("_")[1])
        flags, boost = [0] * 9, [0.0] * 9
        flags[k], boost[k] = 1, 0.5
        m["segment_mode"], m["segment_boost_time"] = flags, boost
    elif name == "image_grid":
        uni["grid_mode"] = 1
        uni["cursor_interaction"] = 1          # must be ignored in grid mode
        uni["mouse_ray_direction"], uni["mouse_ray_origin"] = (0.0, 0.0, 1.0), (64.0, 32.0, 0.0)
        uni["mouse_force"] = (0.5, -0.25, 0.125)
        m.update(grid_width=160, grid_height=80, grid_start_x=4.0, grid_start_y=2.0,
                 grid_spacing_x=0.75, grid_spacing_y=0.75)
    elif name == "image_grid_zero_width":
        uni["grid_mode"] = 1
        m.update(grid_width=0, grid_height=0, grid_start_x=4.0, grid_start_y=2.0,
                 grid_spacing_x=0.75, grid_spacing_y=0.75)
    elif name == "hide_30":
        uni["hide_percentage"] = 30.0
    elif name == "hide_60_sphere_mouse":
        uni["hide_percentage"] = 60.0
        uni["cursor_interaction"] = 1
        uni["mouse_ray_direction"], uni["mouse_ray_origin"] = (0.0, 0.0, 1.0), (64.0, 32.0, 0.0)
        uni["mouse_force"] = (0.5, -0.25, 0.125)
        m["sphere_containment_mode"] = 1
    elif name == "hide_100_front_gravity_segment":
        uni["hide_percentage"] = 100.0
        uni["front_gravity_mode"] = 1
        m["segment_mode"], m["segment_boost_time"] = [0, 1, 0, 0, 0, 0, 0, 0, 0], [0.0, 0.7] + [0.0] * 7
    elif name == "everything":
        uni["cursor_interaction"] = 1
        uni["mouse_ray_direction"], uni["mouse_ray_origin"] = (0.0, 0.0, 1.0), (64.0, 32.0, 0.0)
        uni["mouse_force"] = (0.5, -0.25, 0.125)
        m.update(cylinder_mode=1, sphere_containment_mode=1, two_color_sphere_mode=1,
                 polygon_containment_mode=1, color_spheres_mode=1, color_spheres_boost_time=1.2,
                 segment_mode=[1] * 9, segment_boost_time=[0.1 * k for k in range(9)])
        m["polygon_vertices"], m["polygon_vertex_count"] = _polygon()
        m["color_sphere_positions"], m["palette_colors"] = _color_spheres()
    elif name != "none":
        raise KeyError(name)
    return uni, m


ALL_CASES = (["none", "front_gravity", "front_gravity_cylinder", "sphere", "two_color_sphere", "polygon",
              "polygon_16", "polygon_empty", "color_spheres"] + ["segment_%d" % k for k in range(9)] +
             ["image_grid", "image_grid_zero_width", "everything", "hide_30", "hide_60_sphere_mouse",
              "hide_100_front_gravity_segment"])


def fill(struct_type, fields):
    s = struct_type()
    for key, value in fields.items():
        target = getattr(s, key)
        if isinstance(value, (int, float, np.floating, np.integer)):
            setattr(s, key, value)
        else:
            rows = list(value)
            if rows and isinstance(rows[0], (tuple, list, np.ndarray)):
                for i, row in enumerate(rows):
                    target[i][:] = [float(x) for x in row]
            else:
                target[:] = rows
    return s


def particles(n, seed=3, vel=0.6):
    """Particles all over the box (the default scene is a small blob the containment modes
    would hardly touch), palette colours, moving."""
    from helpers import moving_particles
    rec = moving_particles(n, GRID, seed=seed, vel=vel)
    rng = np.random.default_rng(seed + 100)
    rec[:, 0] = rng.uniform(3.0, GRID[0] - 3.0, n).astype(np.float32)
    rec[:, 1] = rng.uniform(3.0, GRID[1] - 3.0, n).astype(np.float32)
    rec[:, 2] = rng.uniform(3.0, GRID[2] - 3.0, n).astype(np.float32)
    rgb = palette_rgb().astype(np.float32)
    rec[:, 24:27] = rgb[np.arange(n) % rgb.shape[0]]
    return rec
