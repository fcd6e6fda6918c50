"""ctypes binding of the C ABI in include/flowmpm.h (flow_amd/lib/libflowmpm.so).

There is no CPU fallback: a missing library raises at import of the binding, and
a missing GPU makes every computing call fail with FLOWMPM_ERR_HIP.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLOWMPM_LIBRARY points experiments at an alternative build of the same library.
LIB_PATH = os.environ.get("FLOWMPM_LIBRARY") or os.path.join(_HERE, "lib", "libflowmpm.so")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_CAPACITY = 0, 1, 2, 3, 4
PARTICLE_STRIDE = 28
KERNEL_CLEAR_GRID, KERNEL_P2G1, KERNEL_P2G2, KERNEL_UPDATE_GRID, KERNEL_G2P = range(5)
KERNEL_NAMES = ("clearGrid", "p2g1", "p2g2", "updateGrid", "g2p")
TIMING_REBIN, TIMING_REBINS_DONE, TIMING_STRAGGLERS, TIMING_SLOTS = 5, 6, 7, 8
ALGO_SCATTER, ALGO_TILED, ALGO_TILED_UNFUSED = 0, 1, 2
KERNEL_P2G_MASS, KERNEL_P2G_MOMENTUM = 16, 17
KERNEL_CLEAR_FIXED_GRID = 18     # clearGrid without the float grid (updateGrid then keeps it)
# slab handles: the mass pass split around particle migration (include/flowmpm.h)
KERNEL_P2G_MASS_STAYERS, KERNEL_P2G_MASS_ARRIVALS = 19, 20
# slab handles: momentum pass and g2p in an interior and a boundary launch, updateGrid again over the
# boundary tiles (include/flowmpm.h)
KERNEL_P2G_MOMENTUM_INTERIOR, KERNEL_P2G_MOMENTUM_BOUNDARY = 21, 22
KERNEL_G2P_INTERIOR, KERNEL_G2P_BOUNDARY, KERNEL_UPDATE_GRID_BOUNDARY = 23, 24, 25
FIELD_POSITION, FIELD_VELOCITY, FIELD_DENSITY, FIELD_COLOR, FIELD_DIRECTION, FIELD_MASS = range(6)
FIELD_WIDTH = {FIELD_POSITION: 3, FIELD_VELOCITY: 3, FIELD_DENSITY: 1, FIELD_COLOR: 3,
               FIELD_DIRECTION: 3, FIELD_MASS: 1}
PLANE_LOW_HALO, PLANE_LOW_OWNED, PLANE_HIGH_OWNED, PLANE_HIGH_HALO = range(4)
COMP_MOMENTUM, COMP_MASS, COMP_ALL = 0x7, 0x8, 0xF
SIDE_LOW, SIDE_HIGH = 0, 1
MIGRATE_WORDS = 24
SEGMENTS = 9
# Order of the population-segment entries in ModeUniforms (SIM:761-1313).
SEGMENT_NAMES = ("Pink", "Purple", "Cyan", "Teal", "DarkBlue", "DarkPurple", "DarkRed", "Orange", "LightGreen")

# Every symbol include/flowmpm.h declares.
EXPORTS = (
    "flowmpm_version", "flowmpm_device_count", "flowmpm_create", "flowmpm_destroy",
    "flowmpm_last_error", "flowmpm_upload_particles", "flowmpm_set_num_particles",
    "flowmpm_get_num_particles", "flowmpm_set_uniforms", "flowmpm_step", "flowmpm_run_kernel",
    "flowmpm_synchronize", "flowmpm_download_particles", "flowmpm_download_field",
    "flowmpm_get_local_grid", "flowmpm_download_grid_i32", "flowmpm_download_grid_f32",
    "flowmpm_set_profiling", "flowmpm_get_timings", "flowmpm_reset_timings", "flowmpm_plane_bytes",
    "flowmpm_plane_add_i32", "flowmpm_plane_copy_out", "flowmpm_extract_leavers",
    "flowmpm_insert_arrivals", "flowmpm_set_mode_uniforms", "flowmpm_halo_words", "flowmpm_halo_pack",
    "flowmpm_halo_add", "flowmpm_find_leavers", "flowmpm_pack_leavers", "flowmpm_grid_sums",
    "flowmpm_set_domain_particles", "flowmpm_get_allocation_count",
    "flowmpm_download_field_static", "flowmpm_generate_particles", "flowmpm_plane_histogram",
    "flowmpm_stencil_mass_sum", "flowmpm_abi_sizes", "flowmpm_scene_plane_histogram", "flowmpm_set_option",
    "flowmpm_slab_step_begin", "flowmpm_slab_step_pack_mass", "flowmpm_slab_step_interior",
    "flowmpm_slab_step_boundary", "flowmpm_slab_step_grid", "flowmpm_slab_step_end",
    "flowmpm_slab_group_create", "flowmpm_slab_group_destroy", "flowmpm_slab_group_last_error",
    "flowmpm_slab_group_size", "flowmpm_slab_group_slab", "flowmpm_slab_group_upload",
    "flowmpm_slab_group_generate", "flowmpm_slab_group_set_uniforms", "flowmpm_slab_group_set_mode_uniforms",
    "flowmpm_slab_group_step", "flowmpm_slab_group_synchronize", "flowmpm_slab_group_get_num_particles",
    "flowmpm_slab_group_download_particles", "flowmpm_slab_group_download_field",
    "flowmpm_slab_group_conservation", "flowmpm_slab_group_migration_stats", "flowmpm_rccl_selftest",
    "flowmpm_set_slab", "flowmpm_slab_group_rebalance", "flowmpm_slab_group_plane_starts",
)
SCENE_ELLIPSOID, SCENE_DAM_BREAK = 0, 1
OPTION_RIDE_REBINS, OPTION_DRIFT_DIVISOR, OPTION_CHUNK, OPTION_SPARSE_TILE = range(4)


class Config(ctypes.Structure):
    _fields_ = [
        ("grid_x", ctypes.c_int32), ("grid_y", ctypes.c_int32), ("grid_z", ctypes.c_int32),
        ("slab_x0", ctypes.c_int32), ("slab_nx", ctypes.c_int32),
        ("device", ctypes.c_int32), ("algorithm", ctypes.c_int32),
        ("rebin_interval", ctypes.c_int32),
        ("max_particles", ctypes.c_int64),
        ("stream", ctypes.c_void_p),
        ("external_stream", ctypes.c_int32),
        ("slab_max_nx", ctypes.c_int32),
    ]


class Uniforms(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_float), ("stiffness", ctypes.c_float), ("rest_density", ctypes.c_float),
        ("dynamic_viscosity", ctypes.c_float), ("noise", ctypes.c_float), ("time", ctypes.c_float),
        ("mouse_ray_direction", ctypes.c_float * 3),
        ("mouse_ray_origin", ctypes.c_float * 3),
        ("mouse_force", ctypes.c_float * 3),
        ("cursor_interaction", ctypes.c_uint32),
        ("hide_percentage", ctypes.c_float),
        ("grid_mode", ctypes.c_uint32),
        ("front_gravity_mode", ctypes.c_uint32),
    ]


class ModeUniforms(ctypes.Structure):
    """flowmpm_mode_uniforms."""

    _fields_ = [
        ("cylinder_mode", ctypes.c_uint32), ("sphere_containment_mode", ctypes.c_uint32),
        ("polygon_containment_mode", ctypes.c_uint32), ("color_spheres_mode", ctypes.c_uint32),
        ("two_color_sphere_mode", ctypes.c_uint32),
        ("segment_mode", ctypes.c_uint32 * SEGMENTS),
        ("cylinder_radius", ctypes.c_float), ("cylinder_center_x", ctypes.c_float),
        ("cylinder_center_y", ctypes.c_float),
        ("sphere_radius", ctypes.c_float), ("sphere_center", ctypes.c_float * 3),
        ("polygon_vertex_count", ctypes.c_uint32),
        ("polygon_vertices", (ctypes.c_float * 2) * 16),
        ("color_spheres_boost_time", ctypes.c_float), ("color_spheres_boost_duration", ctypes.c_float),
        ("color_sphere_radius", ctypes.c_float),
        ("color_sphere_positions", (ctypes.c_float * 3) * 13),
        ("palette_colors", (ctypes.c_float * 3) * 13),
        ("segment_boost_time", ctypes.c_float * SEGMENTS),
        ("segment_boost_duration", ctypes.c_float * SEGMENTS),
        ("segment_main_radius", ctypes.c_float * SEGMENTS),
        ("segment_sphere_radius", ctypes.c_float * SEGMENTS),
        ("segment_sphere_center", (ctypes.c_float * 3) * SEGMENTS),
        ("segment_negative_center", (ctypes.c_float * 3) * SEGMENTS),
        ("negative_space_sphere_radius", ctypes.c_float),
        ("grid_width", ctypes.c_uint32), ("grid_height", ctypes.c_uint32),
        ("grid_start_x", ctypes.c_float), ("grid_start_y", ctypes.c_float),
        ("grid_spacing_x", ctypes.c_float), ("grid_spacing_y", ctypes.c_float),
        ("grid_z", ctypes.c_float), ("grid_random_offset", ctypes.c_float),
        ("grid_noise_strength", ctypes.c_float),
    ]


class Timings(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double * TIMING_SLOTS), ("launches", ctypes.c_int64 * TIMING_SLOTS)]


class FlowMpmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("flowmpm error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Loads libflowmpm.so once and declares every prototype.  Raises if the
    library has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "flow_amd: %s is missing -- build it with __graft_entry__.build(); "
            "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    i64p = ctypes.POINTER(ctypes.c_int64)
    protos = {
        "flowmpm_version": ([], ctypes.c_int),
        "flowmpm_device_count": ([ctypes.POINTER(ctypes.c_int)], ctypes.c_int),
        "flowmpm_create": ([ctypes.POINTER(Config), ctypes.POINTER(vp)], ctypes.c_int),
        "flowmpm_destroy": ([vp], ctypes.c_int),
        "flowmpm_last_error": ([vp], ctypes.c_char_p),
        "flowmpm_upload_particles": ([vp, fp, ip, i64], ctypes.c_int),
        "flowmpm_set_num_particles": ([vp, i64], ctypes.c_int),
        "flowmpm_get_num_particles": ([vp, i64p, i64p], ctypes.c_int),
        "flowmpm_set_domain_particles": ([vp, i64], ctypes.c_int),
        "flowmpm_set_uniforms": ([vp, ctypes.POINTER(Uniforms)], ctypes.c_int),
        "flowmpm_step": ([vp, i32, ctypes.c_double], ctypes.c_int),
        "flowmpm_run_kernel": ([vp, i32], ctypes.c_int),
        "flowmpm_synchronize": ([vp], ctypes.c_int),
        "flowmpm_download_particles": ([vp, fp, ip, i64, i64p], ctypes.c_int),
        "flowmpm_download_field": ([vp, i32, fp, i64], ctypes.c_int),
        "flowmpm_get_local_grid": ([vp, ip, ip], ctypes.c_int),
        "flowmpm_download_grid_i32": ([vp, ip, i64], ctypes.c_int),
        "flowmpm_download_grid_f32": ([vp, fp, i64], ctypes.c_int),
        "flowmpm_set_profiling": ([vp, i32], ctypes.c_int),
        "flowmpm_get_timings": ([vp, ctypes.POINTER(Timings)], ctypes.c_int),
        "flowmpm_reset_timings": ([vp], ctypes.c_int),
        "flowmpm_plane_bytes": ([vp, i64p], ctypes.c_int),
        "flowmpm_plane_add_i32": ([vp, i32, vp, u32], ctypes.c_int),
        "flowmpm_plane_copy_out": ([vp, i32, vp], ctypes.c_int),
        "flowmpm_extract_leavers": ([vp, ip, i32, vp, i64, i64p], ctypes.c_int),
        "flowmpm_insert_arrivals": ([vp, vp, i64], ctypes.c_int),
        "flowmpm_set_mode_uniforms": ([vp, ctypes.POINTER(ModeUniforms)], ctypes.c_int),
        "flowmpm_halo_words": ([vp, u32, i64p], ctypes.c_int),
        "flowmpm_halo_pack": ([vp, i32, u32, vp], ctypes.c_int),
        "flowmpm_halo_add": ([vp, i32, u32, vp], ctypes.c_int),
        "flowmpm_grid_sums": ([vp, i64p], ctypes.c_int),
        "flowmpm_find_leavers": ([vp, ip, i32, vp], ctypes.c_int),
        "flowmpm_pack_leavers": ([vp, i64p, i32, vp, i64], ctypes.c_int),
        "flowmpm_get_allocation_count": ([vp, i64p], ctypes.c_int),
        "flowmpm_download_field_static": ([vp, i32, fp, i64], ctypes.c_int),
        "flowmpm_generate_particles": ([vp, i32, u32, i64, i64, ctypes.c_double, i64p], ctypes.c_int),
        "flowmpm_plane_histogram": ([vp, i64p, i32], ctypes.c_int),
        "flowmpm_stencil_mass_sum": ([vp, i64p], ctypes.c_int),
        "flowmpm_abi_sizes": ([i64p], ctypes.c_int),
        "flowmpm_set_option": ([vp, i32, i64], ctypes.c_int),
        "flowmpm_slab_step_begin": ([vp], ctypes.c_int),
        "flowmpm_slab_step_pack_mass": ([vp, vp, vp], ctypes.c_int),
        "flowmpm_slab_step_interior": ([vp], ctypes.c_int),
        "flowmpm_slab_step_boundary": ([vp, vp, vp, vp, vp], ctypes.c_int),
        "flowmpm_slab_step_grid": ([vp], ctypes.c_int),
        "flowmpm_slab_step_end": ([vp, vp, vp, ip, i32, vp], ctypes.c_int),
        "flowmpm_slab_group_create": ([ctypes.POINTER(Config), ip, i32, ip, i64p, ctypes.POINTER(vp)], ctypes.c_int),
        "flowmpm_slab_group_destroy": ([vp], ctypes.c_int),
        "flowmpm_slab_group_last_error": ([vp], ctypes.c_char_p),
        "flowmpm_slab_group_size": ([vp, ip, ip], ctypes.c_int),
        "flowmpm_slab_group_slab": ([vp, i32, ctypes.POINTER(vp)], ctypes.c_int),
        "flowmpm_slab_group_upload": ([vp, fp, i64], ctypes.c_int),
        "flowmpm_slab_group_generate": ([vp, i32, u32, i64, ctypes.c_double], ctypes.c_int),
        "flowmpm_slab_group_set_uniforms": ([vp, ctypes.POINTER(Uniforms)], ctypes.c_int),
        "flowmpm_slab_group_set_mode_uniforms": ([vp, ctypes.POINTER(ModeUniforms)], ctypes.c_int),
        "flowmpm_slab_group_step": ([vp, i32, ctypes.c_double], ctypes.c_int),
        "flowmpm_slab_group_synchronize": ([vp], ctypes.c_int),
        "flowmpm_slab_group_get_num_particles": ([vp, i64p, i64p], ctypes.c_int),
        "flowmpm_slab_group_download_particles": ([vp, fp, i64, i64p], ctypes.c_int),
        "flowmpm_slab_group_download_field": ([vp, i32, fp, i64], ctypes.c_int),
        "flowmpm_slab_group_conservation": ([vp, i64p], ctypes.c_int),
        "flowmpm_slab_group_migration_stats": ([vp, i64p, ip], ctypes.c_int),
        "flowmpm_rccl_selftest": ([i32, i64], ctypes.c_int),
        "flowmpm_set_slab": ([vp, i32, i32], ctypes.c_int),
        "flowmpm_slab_group_rebalance": ([vp, ctypes.c_double, ip], ctypes.c_int),
        "flowmpm_slab_group_plane_starts": ([vp, ip], ctypes.c_int),
        "flowmpm_scene_plane_histogram": ([i32, i32, u32, i64, i64, ip, ctypes.c_double, i64p], ctypes.c_int),
    }
    for name, (argtypes, restype) in protos.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = restype
    # the structs above are re-declarations of include/flowmpm.h: their sizes must be the library's
    sizes = (ctypes.c_int64 * 4)()
    L.flowmpm_abi_sizes(sizes)
    mine = [ctypes.sizeof(t) for t in (Config, Uniforms, ModeUniforms, Timings)]
    if list(sizes) != mine:
        raise ImportError("flow_amd: struct layout mismatch with %s: library %r, binding %r"
                          % (LIB_PATH, list(sizes), mine))
    _lib = L
    return L


def check(handle, code):
    if code != OK:
        msg = load().flowmpm_last_error(handle)
        raise FlowMpmError(code, msg.decode() if msg else "")


def check_group(group, code):
    if code != OK:
        msg = load().flowmpm_slab_group_last_error(group)
        raise FlowMpmError(code, msg.decode() if msg else "")


def scene_plane_histogram(scene, count, grid, seed=0xF10E5EED, first_id=0, fill=0.25, device=0):
    """Particles per x-plane of a generated scene (flowmpm_scene_plane_histogram), int64[grid[0]]."""
    import numpy as np
    g = (ctypes.c_int32 * 3)(*[int(x) for x in grid])
    out = np.zeros(int(grid[0]), dtype=np.int64)
    check(None, load().flowmpm_scene_plane_histogram(int(device), int(scene), int(seed) & 0xFFFFFFFF, int(first_id),
                                                     int(count), g, float(fill),
                                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
    return out
