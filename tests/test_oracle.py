"""CPU tests of the oracle itself (no GPU): the C restatement against the independent numpy
restatement, against the committed golden vectors, and against properties that hold for any
conforming execution of the reference's kernels (the reference ships no fixtures)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import DEFAULT_UNIFORMS, assert_bits_equal, moving_particles
from flow_amd.particles import reference_scene
from oracle import mpm_numpy

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_c_oracle_matches_numpy_restatement(oracle_binding):
    grid, n = (24, 16, 12), 1500
    a = moving_particles(n, grid, seed=5)
    b = a.copy()
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    cells = grid[0] * grid[1] * grid[2]
    cell, cellf = np.zeros((cells, 4), np.int32), np.zeros((cells, 4), np.float32)
    for step in range(3):
        orc.step(a, u)
        mpm_numpy.step(b, cell, cellf, grid, n, **{k: v for k, v in DEFAULT_UNIFORMS.items()
                                                   if k in ("dt", "stiffness", "rest_density", "noise", "time")},
                       viscosity=DEFAULT_UNIFORMS["dynamic_viscosity"])
        assert np.array_equal(orc.cell, cell), "fixed-point grid, step %d" % step
        assert_bits_equal(orc.cellf, cellf, "float grid, step %d" % step)
        assert_bits_equal(a, b, "particles, step %d" % step)


def test_c_oracle_matches_numpy_with_mouse_force(oracle_binding):
    grid, n = (16, 16, 16), 400
    a = moving_particles(n, grid, seed=6)
    b = a.copy()
    mouse = dict(mouse_ray_direction=(0.0, 0.6, 0.8), mouse_ray_origin=(8.0, 7.0, 1.0),
                 mouse_force=(0.4, 0.1, -0.2), cursor_interaction=1)
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS, **mouse)
    orc.step(a, u)
    cells = 16 ** 3
    cell, cellf = np.zeros((cells, 4), np.int32), np.zeros((cells, 4), np.float32)
    mpm_numpy.step(b, cell, cellf, grid, n, time=DEFAULT_UNIFORMS["time"], cursor=1,
                   ray_dir=mouse["mouse_ray_direction"], ray_origin=mouse["mouse_ray_origin"],
                   mouse_force=mouse["mouse_force"])
    assert_bits_equal(a, b, "particles with mouse force")


def test_golden_small_case(oracle_binding):
    meta = json.load(open(os.path.join(GOLDEN, "small_16cube_512.json")))
    g = np.load(os.path.join(GOLDEN, "small_16cube_512.npz"))
    grid, n = tuple(meta["grid"]), meta["particles"]
    rec = g["particles_in"].copy()
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n, **meta["uniforms"])
    orc.clear_grid()
    orc.p2g1(rec, u)
    assert np.array_equal(orc.cell, g["cell_after_p2g1"])
    orc.p2g2(rec, u)
    assert np.array_equal(orc.cell, g["cell_after_p2g2"])
    assert_bits_equal(rec[:, 3], g["density_after_p2g2"], "density")
    orc.update_grid(u)
    assert_bits_equal(orc.cellf, g["cellf_after_update"], "cellf")
    orc.g2p(rec, u)
    assert_bits_equal(rec, g["particles_step1"], "particles after step 1")
    orc.run(rec, u, meta["steps_after_first"], meta["dtime"])
    assert_bits_equal(rec, g["particles_step4"], "particles after step 4")
    assert np.array_equal(orc.cell, g["cell_step4"])


def test_golden_default_scene_digest(oracle_binding):
    d = json.load(open(os.path.join(GOLDEN, "default_scene_step1.json")))
    grid, n = tuple(d["grid"]), d["particles"]
    rec = reference_scene(n, grid, seed=d["seed"])
    orc = oracle_binding.Oracle(grid)
    orc.step(rec, oracle_binding.make_uniforms(grid, n, **d["uniforms"]))
    assert hashlib.sha256(orc.cell.tobytes()).hexdigest() == d["sha256_cell_i32"]
    assert hashlib.sha256(orc.cellf.tobytes()).hexdigest() == d["sha256_cellf"]
    assert hashlib.sha256(rec.tobytes()).hexdigest() == d["sha256_particles"]


def test_golden_noise_vectors(oracle_binding):
    rows = json.load(open(os.path.join(GOLDEN, "tri_noise.json")))
    for r in rows:
        out = oracle_binding.tri_noise3d(np.array(r["position"], np.float32), r["speed"], r["time"])
        assert [int(x) for x in out.view(np.uint32)] == r["out_bits"]
        twin = mpm_numpy.tri_noise3d(np.array([r["position"]], np.float32), r["speed"], r["time"])[0]
        assert_bits_equal(out, twin, "noise twin")


def test_rest_state_p2g1_is_pure_mass(oracle_binding):
    """From the initial state (v = 0, C = 0) p2g1 deposits exactly zero momentum and a
    grid mass of sum(trunc(weight * 1e7)): only IEEE multiply/subtract and truncation are
    involved, so this holds for any conforming execution of the reference's kernel."""
    grid, n = (32, 16, 16), 3000
    rec = reference_scene(n, grid, seed=2)
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n)
    orc.clear_grid()
    orc.p2g1(rec, u)
    assert not orc.cell[:, 0:3].any()
    total = int(orc.cell[:, 3].astype(np.int64).sum())
    # each of the 27 truncations loses < 1 unit; the weights sum to 1 up to fp32 rounding
    assert n * 10_000_000 - 30 * n <= total <= n * 10_000_000 + 30 * n
    # and the sum equals the independently computed per-particle truncations
    pos = rec[:, 0:3]
    d = (pos - np.floor(pos)) - np.float32(0.5)
    w = [(np.float32(0.5) * (np.float32(0.5) - d)) * (np.float32(0.5) - d), np.float32(0.75) - d * d,
         (np.float32(0.5) * (np.float32(0.5) + d)) * (np.float32(0.5) + d)]
    want = 0
    for gx in range(3):
        for gy in range(3):
            for gz in range(3):
                wt = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                want += int(np.trunc((wt * np.float32(1e7)).astype(np.float64)).sum())
    assert total == want


def test_grid_is_invariant_under_particle_permutation(oracle_binding):
    """The grid is a sum of integers, so particle order cannot change it."""
    grid, n = (16, 16, 16), 1000
    rec = moving_particles(n, grid, seed=9)
    perm = np.random.default_rng(1).permutation(n)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    a, b = oracle_binding.Oracle(grid), oracle_binding.Oracle(grid)
    ra, rb = rec.copy(), np.ascontiguousarray(rec[perm])
    a.step(ra, u)
    b.step(rb, u)
    assert np.array_equal(a.cell, b.cell)
    assert_bits_equal(ra[perm], rb, "particles under permutation")


def test_openmp_mode_equals_scalar_mode(oracle_binding):
    """The multi-threaded mode used as the CPU baseline computes the same bits."""
    grid, n = (32, 16, 16), 5000
    rec = moving_particles(n, grid, seed=4)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    a, b = oracle_binding.Oracle(grid, threads=1), oracle_binding.Oracle(grid, threads=4)
    ra, rb = rec.copy(), rec.copy()
    a.run(ra, u, 3, 1 / 60)
    b.run(rb, u, 3, 1 / 60)
    assert np.array_equal(a.cell, b.cell)
    assert_bits_equal(ra, rb, "particles, 4 threads vs 1")


def test_fixed_point_wraps_like_i32_atomics(oracle_binding):
    """More than 214.7 units of mass in one cell wraps modulo 2^32 (SIM:43, 326-329)."""
    grid, n = (8, 8, 8), 600
    rec = np.zeros((n, 28), np.float32)
    rec[:, 0:3] = 4.5          # every particle at a cell centre: w1^3 = 0.421875 on cell (4,4,4)
    rec[:, 7] = 1.0
    orc = oracle_binding.Oracle(grid)
    orc.clear_grid()
    orc.p2g1(rec, oracle_binding.make_uniforms(grid, n))
    centre = (4 * 8 + 4) * 8 + 4
    per = int(np.trunc(np.float64(np.float32(0.421875) * np.float32(1e7))))
    want = (n * per + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert n * per > 2 ** 31            # the case really overflows
    assert int(orc.cell[centre, 3]) == want


@pytest.mark.parametrize("x,want", [(2.5e9, 2147483647), (-2.5e9, -2147483648), (float("nan"), 0), (-7.9, -7)])
def test_f32_to_i32_semantics(x, want):
    assert int(mpm_numpy._f2i(np.array([x], np.float32))[0]) == want
