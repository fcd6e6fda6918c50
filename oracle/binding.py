"""ctypes binding of the CPU oracle (oracle/mpm_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (flow_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpm_oracle.so")

STRIDE = 28
POS, DENSITY, VEL, MASS, C, DIR, COLOR = 0, 3, 4, 7, 8, 20, 24


class Uniforms(ctypes.Structure):
    """Mirror of orc_uniforms (oracle/mpm_oracle.h)."""

    _fields_ = [
        ("grid_x", ctypes.c_int32),
        ("grid_y", ctypes.c_int32),
        ("grid_z", ctypes.c_int32),
        ("num_particles", ctypes.c_uint32),
        ("dt", ctypes.c_float),
        ("stiffness", ctypes.c_float),
        ("rest_density", ctypes.c_float),
        ("dynamic_viscosity", ctypes.c_float),
        ("noise", ctypes.c_float),
        ("time", ctypes.c_float),
        ("mouse_ray_direction", ctypes.c_float * 3),
        ("mouse_ray_origin", ctypes.c_float * 3),
        ("mouse_force", ctypes.c_float * 3),
        ("cursor_interaction", ctypes.c_uint32),
        ("hide_percentage", ctypes.c_float),
        ("grid_mode", ctypes.c_uint32),
        ("front_gravity_mode", ctypes.c_uint32),
    ]


SEGMENTS = 9


class ModeUniforms(ctypes.Structure):
    """Mirror of orc_mode_uniforms (oracle/mpm_oracle.h)."""

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


def build(force=False):
    """Compile the oracle with gcc (a no-op when the .so is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in ("mpm_oracle.c", "mpm_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s", "-B", "libmpm_oracle.so"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        up = ctypes.POINTER(Uniforms)
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_get_threads.restype = ctypes.c_int
        L.orc_clear_grid.argtypes = [ip, fp, ctypes.c_int64]
        L.orc_p2g1.argtypes = [fp, ip, up]
        L.orc_p2g2.argtypes = [fp, ip, up]
        L.orc_update_grid.argtypes = [ip, fp, up]
        L.orc_g2p.argtypes = [fp, fp, up]
        L.orc_step.argtypes = [fp, ip, fp, up]
        mp = ctypes.POINTER(ModeUniforms)
        L.orc_g2p_modes.argtypes = [fp, fp, up, mp]
        L.orc_g2p_modes_ids.argtypes = [fp, ip, ctypes.c_int64, fp, up, mp]
        L.orc_step_modes.argtypes = [fp, ip, fp, up, mp]
        L.orc_run.argtypes = [fp, ip, fp, up, ctypes.c_int, ctypes.c_double]
        L.orc_tri_noise3d.argtypes = [fp, ctypes.c_float, ctypes.c_float, fp]
        _lib = L
    return _lib


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def make_uniforms(grid, num_particles, dt=0.1, stiffness=3.0, rest_density=2.5,
                  dynamic_viscosity=0.1, noise=1.0, time=0.0,
                  mouse_ray_direction=(0.0, 0.0, 0.0), mouse_ray_origin=(0.0, 0.0, 0.0),
                  mouse_force=(0.0, 0.0, 0.0), cursor_interaction=0, grid_mode=0, front_gravity_mode=0,
                  hide_percentage=0.0):
    u = Uniforms()
    u.grid_x, u.grid_y, u.grid_z = (int(g) for g in grid)
    u.num_particles = int(num_particles)
    u.dt, u.stiffness, u.rest_density = dt, stiffness, rest_density
    u.dynamic_viscosity, u.noise, u.time = dynamic_viscosity, noise, time
    u.mouse_ray_direction[:] = mouse_ray_direction
    u.mouse_ray_origin[:] = mouse_ray_origin
    u.mouse_force[:] = mouse_force
    u.cursor_interaction = int(cursor_interaction)
    u.hide_percentage = float(hide_percentage)
    u.grid_mode = int(grid_mode)
    u.front_gravity_mode = int(front_gravity_mode)
    return u


class Oracle:
    """Holds the two grids for one problem size and runs oracle kernels on a
    caller-owned AoS particle array (float32, shape [n, 28])."""

    def __init__(self, grid, threads=1):
        self.grid = tuple(int(g) for g in grid)
        self.cell_count = self.grid[0] * self.grid[1] * self.grid[2]
        self.cell = np.zeros((self.cell_count, 4), dtype=np.int32)
        self.cellf = np.zeros((self.cell_count, 4), dtype=np.float32)
        self.L = lib()
        self.threads = threads

    def _t(self):
        self.L.orc_set_threads(self.threads)

    def clear_grid(self):
        self._t()
        self.L.orc_clear_grid(_ip(self.cell), _fp(self.cellf), self.cell_count)

    def p2g1(self, particles, u):
        self._t()
        self.L.orc_p2g1(_fp(particles), _ip(self.cell), ctypes.byref(u))

    def p2g2(self, particles, u):
        self._t()
        self.L.orc_p2g2(_fp(particles), _ip(self.cell), ctypes.byref(u))

    def update_grid(self, u):
        self._t()
        self.L.orc_update_grid(_ip(self.cell), _fp(self.cellf), ctypes.byref(u))

    def g2p(self, particles, u):
        self._t()
        self.L.orc_g2p(_fp(particles), _fp(self.cellf), ctypes.byref(u))

    def step(self, particles, u):
        self._t()
        self.L.orc_step(_fp(particles), _ip(self.cell), _fp(self.cellf), ctypes.byref(u))

    def g2p_modes(self, particles, u, modes, ids=None, count=None):
        """ids: the reference buffer index of each record (default: its position); count: how
        many records `particles` holds when it is a subset of the buffer (u.num_particles is then
        the size of the whole buffer)."""
        self._t()
        if ids is None and count is None:
            self.L.orc_g2p_modes(_fp(particles), _fp(self.cellf), ctypes.byref(u), ctypes.byref(modes))
        else:
            self.L.orc_g2p_modes_ids(_fp(particles), _ip(ids) if ids is not None else None,
                                     ctypes.c_int64(-1 if count is None else int(count)), _fp(self.cellf),
                                     ctypes.byref(u), ctypes.byref(modes))

    def step_modes(self, particles, u, modes):
        self._t()
        self.L.orc_step_modes(_fp(particles), _ip(self.cell), _fp(self.cellf), ctypes.byref(u),
                              ctypes.byref(modes))

    def run(self, particles, u, n_steps, dtime):
        self._t()
        self.L.orc_run(_fp(particles), _ip(self.cell), _fp(self.cellf), ctypes.byref(u),
                       int(n_steps), float(dtime))


def tri_noise3d(pos, speed, time):
    p = np.ascontiguousarray(pos, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_tri_noise3d(_fp(p), speed, time, _fp(out))
    return out
