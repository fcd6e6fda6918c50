"""Host side of the MLS-MPM simulator: the reference's class surface over the HIP library.

`MlsMpmSimulator` mirrors the parts of the reference class that sit on the hot path
(src/mls-mpm/mlsMpmSimulator.js, "SIM"): the constructor (SIM:71-73), init()
(SIM:74-260), update(interval, elapsed) (SIM:2423-2541) and setMouseRay (SIM:1539-1546),
with the same field names (`numParticles`, `gridSize`, `uniforms.<name>.value`,
`fixedPointMultiplier`, `particleBuffer`).  Where the reference hands five kernels to
three.js (SIM:2538-2539) this class calls the C ABI of include/flowmpm.h.

`Handle` is the thin object wrapper over that C ABI which tests and the benchmark
use directly.
"""
import ctypes
import math
import os

import numpy as np

from . import _lib
from .conf import conf as default_conf
from .modes import ModeHost
from .particles import reference_scene
from .structured_array import PARTICLE_STRUCT, StructuredArray


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class Handle:
    """One flowmpm_sim.  Every method maps to one C entry point."""

    def __init__(self, grid, max_particles, device=0, algorithm=_lib.ALGO_TILED, rebin_interval=1,
                 slab=None, stream=None, external_stream=False, slab_max_nx=0):
        self.L = _lib.load()
        cfg = _lib.Config()
        cfg.grid_x, cfg.grid_y, cfg.grid_z = (int(g) for g in grid)
        cfg.slab_x0, cfg.slab_nx = (int(slab[0]), int(slab[1])) if slab else (0, 0)
        cfg.device = int(device)
        cfg.algorithm = int(algorithm)
        cfg.rebin_interval = int(rebin_interval)
        cfg.max_particles = int(max_particles)
        cfg.stream = stream
        cfg.external_stream = 1 if external_stream else 0
        cfg.slab_max_nx = int(slab_max_nx)
        self.grid = (cfg.grid_x, cfg.grid_y, cfg.grid_z)
        self.max_particles = int(max_particles)
        self.slab = slab
        self._h = ctypes.c_void_p()
        _lib.check(None, self.L.flowmpm_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        # A/B switches of the measurement tools: the library itself reads no environment variable
        if algorithm != _lib.ALGO_SCATTER:
            for name, option in (("FLOWMPM_RIDE_REBINS", _lib.OPTION_RIDE_REBINS),
                                 ("FLOWMPM_DRIFT_DIVISOR", _lib.OPTION_DRIFT_DIVISOR),
                                 ("FLOWMPM_CHUNK", _lib.OPTION_CHUNK), ("FLOWMPM_SPARSE_TILE", _lib.OPTION_SPARSE_TILE)):
                value = os.environ.get(name)
                if value is None:
                    continue
                if option == _lib.OPTION_RIDE_REBINS and (slab or algorithm != _lib.ALGO_TILED):
                    continue          # such handles never ride
                self.set_option(option, int(value))

    @classmethod
    def borrowed(cls, pointer, grid, max_particles, slab):
        """A wrapper of a flowmpm_sim somebody else owns (a slab of a flowmpm_slab_group): close() leaves
        it alone."""
        self = cls.__new__(cls)
        self.L = _lib.load()
        self._h = ctypes.c_void_p(pointer)
        self._borrowed = True
        self.grid, self.max_particles, self.slab = tuple(grid), int(max_particles), slab
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.L.flowmpm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, code):
        _lib.check(self._h, code)

    def upload_particles(self, records, ids=None):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, _lib.PARTICLE_STRIDE)
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            assert ids.shape[0] == rec.shape[0]
            idp = _ip(ids)
        self._chk(self.L.flowmpm_upload_particles(self._h, _fp(rec), idp, rec.shape[0]))

    def generate_particles(self, scene, count, seed=0xF10E5EED, first_id=0, fill=0.25):
        """The seeded scene of flow_amd.particles (`reference_scene` / `dam_break`) generated on the
        device, bit-identical to the host records.  Returns the number of particles stored (a slab
        handle keeps those of its own planes)."""
        stored = ctypes.c_int64()
        self._chk(self.L.flowmpm_generate_particles(self._h, int(scene), int(seed) & 0xFFFFFFFF, int(first_id),
                                                    int(count), float(fill), ctypes.byref(stored)))
        return stored.value

    def plane_histogram(self):
        """Resident particles per x-plane of the global grid (int64[grid_x])."""
        out = np.zeros(self.grid[0], dtype=np.int64)
        self._chk(self.L.flowmpm_plane_histogram(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                 self.grid[0]))
        return out

    def stencil_mass_sum(self):
        """Sum of encodeFixedPoint(weight) over the resident particles' stencils, from the positions."""
        out = ctypes.c_int64()
        self._chk(self.L.flowmpm_stencil_mass_sum(self._h, ctypes.byref(out)))
        return out.value

    def set_option(self, option, value):
        """flowmpm_set_option: tuning of the tiled algorithms (never affects results)."""
        self._chk(self.L.flowmpm_set_option(self._h, int(option), int(value)))

    def set_num_particles(self, n):
        self._chk(self.L.flowmpm_set_num_particles(self._h, int(n)))

    def set_domain_particles(self, n):
        """Slab handles: uniforms.numParticles of the whole domain (the hide-percentage branch
        divides the particle id by it)."""
        self._chk(self.L.flowmpm_set_domain_particles(self._h, int(n)))

    def num_particles(self):
        live, resident = ctypes.c_int64(), ctypes.c_int64()
        self._chk(self.L.flowmpm_get_num_particles(self._h, ctypes.byref(live), ctypes.byref(resident)))
        return live.value, resident.value

    def set_uniforms(self, dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0,
                     time=0.0, mouse_ray_direction=(0.0, 0.0, 0.0), mouse_ray_origin=(0.0, 0.0, 0.0),
                     mouse_force=(0.0, 0.0, 0.0), cursor_interaction=0, hide_percentage=0.0,
                     grid_mode=0, front_gravity_mode=0):
        u = _lib.Uniforms()
        u.dt, u.stiffness, u.rest_density = dt, stiffness, rest_density
        u.dynamic_viscosity, u.noise, u.time = dynamic_viscosity, noise, time
        u.mouse_ray_direction[:] = mouse_ray_direction
        u.mouse_ray_origin[:] = mouse_ray_origin
        u.mouse_force[:] = mouse_force
        u.cursor_interaction = int(cursor_interaction)
        u.hide_percentage = hide_percentage
        u.grid_mode, u.front_gravity_mode = int(grid_mode), int(front_gravity_mode)
        self._chk(self.L.flowmpm_set_uniforms(self._h, ctypes.byref(u)))

    def set_mode_uniforms(self, modes):
        """`modes` is a _lib.ModeUniforms record."""
        self._chk(self.L.flowmpm_set_mode_uniforms(self._h, ctypes.byref(modes)))

    def step(self, n_steps=1, dtime=0.0):
        self._chk(self.L.flowmpm_step(self._h, int(n_steps), float(dtime)))

    def run_kernel(self, kernel_id):
        self._chk(self.L.flowmpm_run_kernel(self._h, int(kernel_id)))

    def synchronize(self):
        self._chk(self.L.flowmpm_synchronize(self._h))

    def download_particles(self):
        """Records in particle-id order (single GPU), or (records, ids) for a slab."""
        _, resident = self.num_particles()
        rec = np.zeros((max(resident, 1), _lib.PARTICLE_STRIDE), dtype=np.float32)
        ids = np.zeros(max(resident, 1), dtype=np.int32)
        count = ctypes.c_int64()
        self._chk(self.L.flowmpm_download_particles(self._h, _fp(rec), _ip(ids), rec.shape[0],
                                                    ctypes.byref(count)))
        if self.slab:
            return rec[:count.value], ids[:count.value]
        return rec[:count.value]

    def download_field(self, field):
        live, _ = self.num_particles()
        out = np.zeros((max(live, 1), _lib.FIELD_WIDTH[field]), dtype=np.float32)
        self._chk(self.L.flowmpm_download_field(self._h, int(field), _fp(out), out.shape[0]))
        return out[:live]

    def download_field_static(self, field):
        """download_field through the library's static device array: no allocation, at most
        262,144 particles."""
        live, _ = self.num_particles()
        out = np.zeros((max(live, 1), _lib.FIELD_WIDTH[field]), dtype=np.float32)
        self._chk(self.L.flowmpm_download_field_static(self._h, int(field), _fp(out), out.shape[0]))
        return out[:live]

    def local_grid(self):
        x0, nx = ctypes.c_int32(), ctypes.c_int32()
        self._chk(self.L.flowmpm_get_local_grid(self._h, ctypes.byref(x0), ctypes.byref(nx)))
        return x0.value, nx.value

    def download_grid_i32(self):
        _, nx = self.local_grid()
        cells = nx * self.grid[1] * self.grid[2]
        out = np.zeros((cells, 4), dtype=np.int32)
        self._chk(self.L.flowmpm_download_grid_i32(self._h, _ip(out), cells))
        return out

    def download_grid_f32(self):
        _, nx = self.local_grid()
        cells = nx * self.grid[1] * self.grid[2]
        out = np.zeros((cells, 4), dtype=np.float32)
        self._chk(self.L.flowmpm_download_grid_f32(self._h, _fp(out), cells))
        return out

    def grid_sums(self):
        """int64 sums of the owned fixed-point grid: [x, y, z, mass]."""
        out = np.zeros(4, dtype=np.int64)
        self._chk(self.L.flowmpm_grid_sums(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return out

    def set_profiling(self, every_nth_step):
        """0 / False: off; 1 / True: every launch; k: the launches of every k-th step()."""
        self._chk(self.L.flowmpm_set_profiling(self._h, int(every_nth_step)))

    def allocation_count(self):
        """Device / pinned allocations the handle has made for staging since it was created."""
        out = ctypes.c_int64(0)
        self._chk(self.L.flowmpm_get_allocation_count(self._h, ctypes.byref(out)))
        return out.value

    def reset_timings(self):
        self._chk(self.L.flowmpm_reset_timings(self._h))

    def timings(self):
        t = _lib.Timings()
        self._chk(self.L.flowmpm_get_timings(self._h, ctypes.byref(t)))
        names = list(_lib.KERNEL_NAMES) + ["rebin"]
        out = {names[i]: {"ms": t.ms[i], "launches": t.launches[i]} for i in range(len(names))}
        out["rebin"]["performed"] = t.launches[_lib.TIMING_REBINS_DONE]
        out["rebin"]["performed_ms"] = t.ms[_lib.TIMING_REBINS_DONE]      # device clock, every step
        out["rebin"]["stragglers"] = t.launches[_lib.TIMING_STRAGGLERS]     # particles, summed over the g2p stages
        return out

    # ---- slab decomposition ----
    def plane_bytes(self):
        nbytes = ctypes.c_int64()
        self._chk(self.L.flowmpm_plane_bytes(self._h, ctypes.byref(nbytes)))
        return nbytes.value

    def plane_add_i32(self, plane, dev_src, comp_mask):
        self._chk(self.L.flowmpm_plane_add_i32(self._h, int(plane), ctypes.c_void_p(dev_src), int(comp_mask)))

    def plane_copy_out(self, plane, dev_dst):
        self._chk(self.L.flowmpm_plane_copy_out(self._h, int(plane), ctypes.c_void_p(dev_dst)))

    def halo_words(self, comp_mask):
        words = ctypes.c_int64()
        self._chk(self.L.flowmpm_halo_words(self._h, int(comp_mask), ctypes.byref(words)))
        return words.value

    def halo_pack(self, side, comp_mask, dev_dst):
        self._chk(self.L.flowmpm_halo_pack(self._h, int(side), int(comp_mask), ctypes.c_void_p(dev_dst)))

    def halo_add(self, side, comp_mask, dev_src):
        self._chk(self.L.flowmpm_halo_add(self._h, int(side), int(comp_mask), ctypes.c_void_p(dev_src)))

    def extract_leavers(self, plane_starts, dev_out, capacity_records):
        starts = np.ascontiguousarray(plane_starts, dtype=np.int32)
        n_ranks = starts.shape[0] - 1
        counts = np.zeros(n_ranks, dtype=np.int64)
        self._chk(self.L.flowmpm_extract_leavers(
            self._h, _ip(starts), n_ranks, ctypes.c_void_p(dev_out), int(capacity_records),
            counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return counts

    def find_leavers(self, plane_starts, dev_counts):
        starts = np.ascontiguousarray(plane_starts, dtype=np.int32)
        self._chk(self.L.flowmpm_find_leavers(self._h, _ip(starts), starts.shape[0] - 1, ctypes.c_void_p(dev_counts)))

    def pack_leavers(self, counts, dev_out, capacity_records):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        self._chk(self.L.flowmpm_pack_leavers(
            self._h, counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), counts.shape[0],
            ctypes.c_void_p(dev_out), int(capacity_records)))

    def set_slab(self, slab_x0, slab_nx):
        """flowmpm_set_slab: re-cuts the handle (between steps; follow with find_leavers under the new cut)."""
        self._chk(self.L.flowmpm_set_slab(self._h, int(slab_x0), int(slab_nx)))
        self.slab = (int(slab_x0), int(slab_nx))

    # one slab step in six calls (include/flowmpm.h); buffers are device addresses, 0 / None = no neighbour
    def slab_step_begin(self):
        self._chk(self.L.flowmpm_slab_step_begin(self._h))

    def slab_step_pack_mass(self, send_low, send_high):
        self._chk(self.L.flowmpm_slab_step_pack_mass(self._h, ctypes.c_void_p(send_low), ctypes.c_void_p(send_high)))

    def slab_step_interior(self):
        self._chk(self.L.flowmpm_slab_step_interior(self._h))

    def slab_step_boundary(self, recv_low, recv_high, send_low, send_high):
        self._chk(self.L.flowmpm_slab_step_boundary(self._h, ctypes.c_void_p(recv_low), ctypes.c_void_p(recv_high),
                                                    ctypes.c_void_p(send_low), ctypes.c_void_p(send_high)))

    def slab_step_grid(self):
        self._chk(self.L.flowmpm_slab_step_grid(self._h))

    def slab_step_end(self, recv_low, recv_high, plane_starts=None, dev_counts=None):
        if plane_starts is None:
            starts, n_ranks = None, 0
        else:
            arr = np.ascontiguousarray(plane_starts, dtype=np.int32)
            starts, n_ranks = _ip(arr), arr.shape[0] - 1
        self._chk(self.L.flowmpm_slab_step_end(self._h, ctypes.c_void_p(recv_low), ctypes.c_void_p(recv_high), starts,
                                               n_ranks, ctypes.c_void_p(dev_counts)))

    def insert_arrivals(self, dev_in, count):
        self._chk(self.L.flowmpm_insert_arrivals(self._h, ctypes.c_void_p(dev_in), int(count)))


class SlabGroup:
    """flowmpm_slab_group: a decomposed domain behind one object, all slabs driven by this thread (on
    one device, or one slab per device over RCCL).  Mirrors Handle where the calls exist for a group."""

    def __init__(self, grid, plane_starts, max_particles, devices=None, rebin_interval=64, slab_max_nx=0):
        self.L = _lib.load()
        self.grid = tuple(int(g) for g in grid)
        self.plane_starts = [int(x) for x in plane_starts]
        self.world = len(self.plane_starts) - 1
        caps = list(max_particles) if hasattr(max_particles, "__len__") else [int(max_particles)] * self.world
        cfg = _lib.Config()
        cfg.grid_x, cfg.grid_y, cfg.grid_z = self.grid
        cfg.algorithm, cfg.rebin_interval = _lib.ALGO_TILED, int(rebin_interval)
        cfg.slab_max_nx = int(slab_max_nx)
        cfg.device = int(devices[0]) if devices else 0
        starts = (ctypes.c_int32 * (self.world + 1))(*self.plane_starts)
        devs = (ctypes.c_int32 * self.world)(*[int(d) for d in devices]) if devices else None
        cap = (ctypes.c_int64 * self.world)(*[int(c) for c in caps])
        self._g = ctypes.c_void_p()
        _lib.check_group(None, self.L.flowmpm_slab_group_create(ctypes.byref(cfg), starts, self.world, devs, cap,
                                                                ctypes.byref(self._g)))
        self.handles = []
        for r in range(self.world):
            h = ctypes.c_void_p()
            self._chk(self.L.flowmpm_slab_group_slab(self._g, r, ctypes.byref(h)))
            self.handles.append(Handle.borrowed(h.value, self.grid, caps[r],
                                                (self.plane_starts[r], self.plane_starts[r + 1] - self.plane_starts[r])))

    def _chk(self, code):
        _lib.check_group(self._g, code)

    def close(self):
        if getattr(self, "_g", None):
            self.L.flowmpm_slab_group_destroy(self._g)
            self._g = None
            self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def uses_rccl(self):
        n, rccl = ctypes.c_int32(), ctypes.c_int32()
        self._chk(self.L.flowmpm_slab_group_size(self._g, ctypes.byref(n), ctypes.byref(rccl)))
        return bool(rccl.value)

    def upload_particles(self, records):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, _lib.PARTICLE_STRIDE)
        self._chk(self.L.flowmpm_slab_group_upload(self._g, _fp(rec), rec.shape[0]))

    def generate_particles(self, scene, count, seed=0xF10E5EED, fill=0.25):
        self._chk(self.L.flowmpm_slab_group_generate(self._g, int(scene), int(seed) & 0xFFFFFFFF, int(count), float(fill)))

    def set_uniforms(self, **kw):
        u = _lib.Uniforms()
        d = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.0,
                 mouse_ray_direction=(0.0, 0.0, 0.0), mouse_ray_origin=(0.0, 0.0, 0.0), mouse_force=(0.0, 0.0, 0.0),
                 cursor_interaction=0, hide_percentage=0.0, grid_mode=0, front_gravity_mode=0)
        d.update(kw)
        for name in ("dt", "stiffness", "rest_density", "dynamic_viscosity", "noise", "time", "hide_percentage"):
            setattr(u, name, d[name])
        u.mouse_ray_direction[:] = d["mouse_ray_direction"]
        u.mouse_ray_origin[:] = d["mouse_ray_origin"]
        u.mouse_force[:] = d["mouse_force"]
        u.cursor_interaction, u.grid_mode, u.front_gravity_mode = (int(d["cursor_interaction"]), int(d["grid_mode"]),
                                                                   int(d["front_gravity_mode"]))
        self._chk(self.L.flowmpm_slab_group_set_uniforms(self._g, ctypes.byref(u)))

    def set_mode_uniforms(self, modes):
        self._chk(self.L.flowmpm_slab_group_set_mode_uniforms(self._g, ctypes.byref(modes)))

    def step(self, n_steps=1, dtime=0.0):
        self._chk(self.L.flowmpm_slab_group_step(self._g, int(n_steps), float(dtime)))

    def synchronize(self):
        self._chk(self.L.flowmpm_slab_group_synchronize(self._g))

    def num_particles(self):
        total = ctypes.c_int64()
        per = np.zeros(self.world, dtype=np.int64)
        self._chk(self.L.flowmpm_slab_group_get_num_particles(self._g, ctypes.byref(total),
                                                              per.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return total.value, per

    def download_particles(self):
        total, _ = self.num_particles()
        rec = np.zeros((max(total, 1), _lib.PARTICLE_STRIDE), dtype=np.float32)
        count = ctypes.c_int64()
        self._chk(self.L.flowmpm_slab_group_download_particles(self._g, _fp(rec), rec.shape[0], ctypes.byref(count)))
        return rec[:count.value]

    def download_field(self, field):
        total, _ = self.num_particles()
        out = np.zeros((max(total, 1), _lib.FIELD_WIDTH[field]), dtype=np.float32)
        self._chk(self.L.flowmpm_slab_group_download_field(self._g, int(field), _fp(out), out.shape[0]))
        return out[:total]

    def rebalance(self, max_move_fraction=0.1):
        """flowmpm_slab_group_rebalance: re-cuts the slabs at the particle quantiles of the moment; True when
        the cut moved (plane_starts and the handles' slabs are refreshed)."""
        changed = ctypes.c_int32()
        self._chk(self.L.flowmpm_slab_group_rebalance(self._g, float(max_move_fraction), ctypes.byref(changed)))
        if changed.value:
            starts = (ctypes.c_int32 * (self.world + 1))()
            self._chk(self.L.flowmpm_slab_group_plane_starts(self._g, starts))
            self.plane_starts = [int(x) for x in starts]
            for r, h in enumerate(self.handles):
                h.slab = (self.plane_starts[r], self.plane_starts[r + 1] - self.plane_starts[r])
        return bool(changed.value)

    def conservation(self):
        out = np.zeros(5, dtype=np.int64)
        self._chk(self.L.flowmpm_slab_group_conservation(self._g, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return out

    def migration_stats(self):
        moved, hop = ctypes.c_int64(), ctypes.c_int32()
        self._chk(self.L.flowmpm_slab_group_migration_stats(self._g, ctypes.byref(moved), ctypes.byref(hop)))
        return moved.value, hop.value


def rccl_selftest(device=0, words=1 << 16):
    """flowmpm_rccl_selftest: a world-of-one RCCL communicator sends a buffer to itself."""
    _lib.check(None, _lib.load().flowmpm_rccl_selftest(int(device), int(words)))


class Vector3:
    """The few three.js Vector3 operations the simulator's host code uses
    (setMouseRay mutates its arguments, SIM:1540-1545)."""

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def set(self, x, y, z):
        self.x, self.y, self.z = float(x), float(y), float(z)
        return self

    def copy(self, v):
        self.x, self.y, self.z = v.x, v.y, v.z
        return self

    def clone(self):
        return Vector3(self.x, self.y, self.z)

    def add(self, v):
        self.x += v.x; self.y += v.y; self.z += v.z
        return self

    def sub(self, v):
        self.x -= v.x; self.y -= v.y; self.z -= v.z
        return self

    def multiplyScalar(self, s):
        self.x *= s; self.y *= s; self.z *= s
        return self

    def divideScalar(self, s):
        return self.multiplyScalar(1 / s)

    def length(self):
        return math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)

    def normalize(self):
        return self.divideScalar(self.length() or 1)

    def toArray(self):
        return [self.x, self.y, self.z]


class Uniform:
    """`uniforms.<name>.value`, as three.js's uniform() nodes expose it."""

    def __init__(self, value):
        self.value = value


class _Uniforms:
    pass


class MlsMpmSimulator:
    fixedPointMultiplier = 1e7          # SIM:43

    def __init__(self, renderer=None, conf=None, device=0, algorithm=_lib.ALGO_TILED, rebin_interval=64,
                 seed=0xF10E5EED, clock=None):
        self.clock = clock               # stands in for Date.now(), milliseconds (SIM:1827, 2428)
        self.renderer = renderer         # kept for signature parity; unused (SIM:71-73)
        self.conf = conf if conf is not None else default_conf
        self.device = device
        self.algorithm = algorithm
        self.rebin_interval = rebin_interval
        self.seed = seed
        self.numParticles = 0
        self.gridSize = Vector3(0, 0, 0)
        self.uniforms = _Uniforms()
        self.mousePos = Vector3()
        self.mousePosArray = []
        self.particleBuffer = None
        self.handle = None

    def init(self):
        """SIM:74-260: grid size, particle records, grids, uniforms."""
        max_particles = self.conf.maxParticles
        self.gridSize.set(128, 64, 64)                                   # SIM:76
        grid = (128, 64, 64)
        self.particleBuffer = StructuredArray(PARTICLE_STRUCT, max_particles, "particleData")
        self.particleBuffer.records()[:] = reference_scene(max_particles, grid, self.seed)  # SIM:114-129
        u = self.uniforms
        u.gravityType = Uniform(0)
        u.gravity = Uniform(Vector3())
        u.stiffness = Uniform(0.0)
        u.restDensity = Uniform(0.0)
        u.dynamicViscosity = Uniform(0.0)
        u.noise = Uniform(0.0)
        u.gridSize = Uniform(self.gridSize)
        u.dt = Uniform(0.1)
        u.numParticles = Uniform(0)
        u.gridMode = Uniform(0)
        u.frontGravityMode = Uniform(0)
        u.mouseRayDirection = Uniform(Vector3())
        u.mouseRayOrigin = Uniform(Vector3())
        u.mouseForce = Uniform(Vector3())
        u.cursorInteraction = Uniform(0)
        u.hidePercentage = Uniform(0.0)
        u.time = Uniform(0.0)            # three.js TSL `time` node (SIM:22, 1318)
        self.modes = ModeHost(u, Uniform, Vector3, clock=self.clock)     # SIM:154-254
        self.handle = Handle(grid, max_particles, device=self.device, algorithm=self.algorithm,
                             rebin_interval=self.rebin_interval)
        self.handle.upload_particles(self.particleBuffer.records())
        return self

    def setMouseRay(self, origin, direction, pos):
        """SIM:1539-1546.  Mutates `origin`, `direction` and `pos` like the reference."""
        origin.multiplyScalar(64)
        pos.multiplyScalar(64)
        origin.add(Vector3(64, 0, 0))
        self.uniforms.mouseRayDirection.value.copy(direction.normalize())
        self.uniforms.mouseRayOrigin.value.copy(origin)
        self.mousePos.copy(pos)

    def toggleGridMode(self, mode, image=None):
        """SIM:1776-2421.  Image modes (1, 2, 7) take the decoded image as an array
        [height, width, >= 3] uint8 or the path of a binary PPM file.  The simulated state and
        the colours g2p reads stay as they are (see modes.py); getColors() returns the
        rewritten colours from now on."""
        colors = self.particleBuffer.records()[:, 24:27]
        self.modes.toggle(int(mode), colors, self.conf.particles, image=image)

    @property
    def gridMode(self):
        return self.modes.gridMode

    def update(self, interval, elapsed):
        """SIM:2423-2541: one frame = one step.  `elapsed` supplies the clock that the
        reference's noise reads from three.js's `time` node."""
        c = self.conf
        u = self.uniforms
        self.modes.update_boost_clocks()                                 # SIM:2426-2494
        u.noise.value = c.noise                                          # SIM:2496-2511
        u.gridNoiseStrength.value = c.gridNoiseStrength
        u.stiffness.value = c.stiffness
        u.gravityType.value = c.gravity
        u.gravity.value.set(0, 0, 0)
        u.cursorInteraction.value = 1 if c.cursorInteraction else 0
        u.hidePercentage.value = c.hidePercentage
        u.dynamicViscosity.value = c.dynamicViscosity
        u.restDensity.value = c.restDensity
        u.time.value = elapsed

        if c.particles != self.numParticles:                             # SIM:2513-2522
            self.numParticles = c.particles
            u.numParticles.value = c.particles
            self.handle.set_num_particles(c.particles)

        interval = min(interval, 1 / 60)                                 # SIM:2524-2528
        effective_speed = 4.0 if u.frontGravityMode.value == 1 else c.speed
        u.dt.value = interval * 6 * effective_speed

        self.mousePosArray.append(self.mousePos.clone())                 # SIM:2530-2534
        if len(self.mousePosArray) > 3:
            self.mousePosArray.pop(0)
        if len(self.mousePosArray) > 1:
            u.mouseForce.value.copy(self.mousePosArray[-1]).sub(self.mousePosArray[0]) \
                .divideScalar(len(self.mousePosArray))

        if c.run:                                                        # SIM:2537-2540
            self._push_uniforms()
            self.handle.step(1, 0.0)

    def _push_uniforms(self):
        u = self.uniforms
        self.handle.set_uniforms(
            dt=u.dt.value, stiffness=u.stiffness.value, rest_density=u.restDensity.value,
            dynamic_viscosity=u.dynamicViscosity.value, noise=u.noise.value, time=u.time.value,
            mouse_ray_direction=u.mouseRayDirection.value.toArray(),
            mouse_ray_origin=u.mouseRayOrigin.value.toArray(),
            mouse_force=u.mouseForce.value.toArray(),
            cursor_interaction=u.cursorInteraction.value, hide_percentage=u.hidePercentage.value,
            grid_mode=u.gridMode.value, front_gravity_mode=u.frontGravityMode.value)
        self.handle.set_mode_uniforms(self.modes.pack())

    # ---- what the reference's renderers bind (particleRenderer.js:119-133,
    # pointRenderer.js:17-19), as host arrays in particle-id order ----
    def getPositions(self):
        return self.handle.download_field(_lib.FIELD_POSITION)

    def getVelocities(self):
        return self.handle.download_field(_lib.FIELD_VELOCITY)

    def getDensities(self):
        return self.handle.download_field(_lib.FIELD_DENSITY)

    def getDirections(self):
        return self.handle.download_field(_lib.FIELD_DIRECTION)

    def getColors(self):
        """What material.colorNode reads: the device colours until toggleGridMode first
        rebinds the material to a rewritten buffer (SIM:2376-2384)."""
        if self.modes.render_colors is not None:
            return self.modes.render_colors[:self.numParticles].copy()
        return self.handle.download_field(_lib.FIELD_COLOR)

    def readAoS(self):
        """All maxParticles records in the reference's 28-float layout; also refreshes
        particleBuffer.floatArray."""
        rec = self.handle.download_particles()
        host = self.particleBuffer.records()
        host[:rec.shape[0], :24] = rec[:, :24]      # colours stay the host's (toggleGridMode rewrites them)
        host[:rec.shape[0], 27] = rec[:, 27]
        return rec

    def synchronize(self):
        self.handle.synchronize()
