"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for the fixed-point grid (integers) AND for the float
grid and particle state (the kernels use the oracle's IEEE operation order with fused
multiply-add disabled; the stated requirement is only 1e-5 relative for particle state)."""
import hashlib
import json
import os

import numpy as np
import pytest

import flow_amd
from flow_amd import _lib
from flow_amd.particles import dam_break, reference_scene
from helpers import DEFAULT_UNIFORMS, assert_bits_equal, moving_particles

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGOS = [pytest.param(_lib.ALGO_SCATTER, id="scatter"), pytest.param(_lib.ALGO_TILED, id="tiled")]


def _handle(grid, rec, algo, uniforms=DEFAULT_UNIFORMS, rebin_interval=1, max_particles=None):
    h = flow_amd.Handle(grid, max_particles or max(rec.shape[0], 1), algorithm=algo,
                        rebin_interval=rebin_interval)
    h.upload_particles(rec)
    h.set_uniforms(**uniforms)
    return h


@pytest.mark.parametrize("algo", ALGOS)
def test_each_kernel_matches_oracle(oracle_binding, algo):
    grid, n = (32, 24, 16), 6000
    rec = moving_particles(n, grid, seed=21)
    h = _handle(grid, rec, algo)
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    want = rec.copy()

    h.run_kernel(_lib.KERNEL_CLEAR_GRID)
    orc.clear_grid()
    assert not h.download_grid_i32().any() and not h.download_grid_f32().any()

    h.run_kernel(_lib.KERNEL_P2G1)
    orc.p2g1(want, u)
    assert np.array_equal(h.download_grid_i32(), orc.cell), "grid after p2g1"

    h.run_kernel(_lib.KERNEL_P2G2)
    orc.p2g2(want, u)
    assert np.array_equal(h.download_grid_i32(), orc.cell), "grid after p2g2"
    assert_bits_equal(h.download_field(_lib.FIELD_DENSITY)[:, 0], want[:, 3], "density after p2g2")

    h.run_kernel(_lib.KERNEL_UPDATE_GRID)
    orc.update_grid(u)
    assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid after updateGrid")

    h.run_kernel(_lib.KERNEL_G2P)
    orc.g2p(want, u)
    assert_bits_equal(h.download_particles(), want, "particles after g2p")
    h.close()


def test_two_pass_p2g_equals_reference_p2g1_plus_p2g2(oracle_binding):
    """The tiled path's mass pass + summed-momentum pass must leave exactly the grid the
    reference has after p2g2 (each term is encoded on its own, then the integers are added)."""
    grid, n = (32, 24, 16), 6000
    rec = moving_particles(n, grid, seed=28, vel=1.0)
    orc = oracle_binding.Oracle(grid)
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    want = rec.copy()
    orc.clear_grid()
    orc.p2g1(want, u)
    mass_after_p2g1 = orc.cell[:, 3].copy()
    orc.p2g2(want, u)
    h = _handle(grid, rec, _lib.ALGO_TILED)
    h.run_kernel(_lib.KERNEL_CLEAR_GRID)
    h.run_kernel(_lib.KERNEL_P2G_MASS)
    got = h.download_grid_i32()
    assert np.array_equal(got[:, 3], mass_after_p2g1) and not got[:, 0:3].any()
    h.run_kernel(_lib.KERNEL_P2G_MOMENTUM)
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_field(_lib.FIELD_DENSITY)[:, 0], want[:, 3], "density")
    h.close()
    # and the A/B algorithm that keeps the reference's staging inside step()
    a, b = _handle(grid, rec, _lib.ALGO_TILED), _handle(grid, rec, _lib.ALGO_TILED_UNFUSED)
    a.step(5, 0.1)
    b.step(5, 0.1)
    assert np.array_equal(a.download_grid_i32(), b.download_grid_i32())
    assert_bits_equal(a.download_particles(), b.download_particles(), "fused vs unfused step")
    a.close()
    b.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_golden_small_case(algo):
    """Committed vectors (tests/golden/make_golden.py), including the mouse-force branch."""
    meta = json.load(open(os.path.join(GOLDEN, "small_16cube_512.json")))
    g = np.load(os.path.join(GOLDEN, "small_16cube_512.npz"))
    grid = tuple(meta["grid"])
    h = _handle(grid, g["particles_in"], algo, uniforms=meta["uniforms"])
    h.run_kernel(_lib.KERNEL_CLEAR_GRID)
    h.run_kernel(_lib.KERNEL_P2G1)
    assert np.array_equal(h.download_grid_i32(), g["cell_after_p2g1"])
    h.run_kernel(_lib.KERNEL_P2G2)
    assert np.array_equal(h.download_grid_i32(), g["cell_after_p2g2"])
    h.run_kernel(_lib.KERNEL_UPDATE_GRID)
    assert_bits_equal(h.download_grid_f32(), g["cellf_after_update"], "cellf")
    h.run_kernel(_lib.KERNEL_G2P)
    assert_bits_equal(h.download_particles(), g["particles_step1"], "particles step 1")
    h.step(meta["steps_after_first"], meta["dtime"])
    assert_bits_equal(h.download_particles(), g["particles_step4"], "particles step 4")
    assert np.array_equal(h.download_grid_i32(), g["cell_step4"])
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_default_scene_digest(algo):
    """The reference's default scene (81,920 particles, 128x64x64), one step."""
    d = json.load(open(os.path.join(GOLDEN, "default_scene_step1.json")))
    grid, n = tuple(d["grid"]), d["particles"]
    h = _handle(grid, reference_scene(n, grid, seed=d["seed"]), algo, uniforms=d["uniforms"])
    h.step(1)
    cell = h.download_grid_i32()
    assert hashlib.sha256(cell.tobytes()).hexdigest() == d["sha256_cell_i32"]
    assert hashlib.sha256(h.download_grid_f32().tobytes()).hexdigest() == d["sha256_cellf"]
    assert hashlib.sha256(h.download_particles().tobytes()).hexdigest() == d["sha256_particles"]
    assert int(cell[:, 3].astype(np.int64).sum()) == d["grid_mass_sum_i64"]
    h.close()


@pytest.mark.parametrize("case", ["fast", "crowded"])
def test_long_run_on_one_binning(oracle_binding, monkeypatch, case):
    """No re-bin after the first one (the drift rule is off and the interval is never reached):
    every particle that leaves its tile's arena goes through the follow-up kernels (32 lanes to a
    particle in the two p2g passes, the momentum pass's list reused by g2p) for as long as the
    run lasts, and must still give the reference's bits."""
    monkeypatch.setenv("FLOWMPM_DRIFT_DIVISOR", "0")
    if case == "fast":
        grid, n, steps = (40, 24, 16), 6000, 60
        rec = moving_particles(n, grid, seed=77, vel=3.0)
    else:
        # a block of fluid that fills a few tiles and spreads into empty ones: thousands of
        # particles end up in tiles that have no work item
        grid, n, steps = (32, 32, 32), 20000, 40
        rec = moving_particles(n, grid, seed=78, vel=2.5)
        rec[:, 0:3] = 10.0 + (rec[:, 0:3] - 2.0) * (9.0 / 28.0)
    h = _handle(grid, rec, _lib.ALGO_TILED, rebin_interval=100000)
    orc = oracle_binding.Oracle(grid, threads=8)
    want = rec.copy()
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    for chunk in range(steps // 10):
        h.step(10, 0.0)
        orc.run(want, u, 10, 0.0)
        assert np.array_equal(h.download_grid_i32(), orc.cell), "fixed-point grid after %d steps" % (10 * chunk + 10)
        assert_bits_equal(h.download_particles(), want, "particles after %d steps" % (10 * chunk + 10))
    t = h.timings()["rebin"]
    assert t["performed"] == 1
    assert t["stragglers"] > n, "the case must send particles through the follow-up kernels"
    h.close()


def test_g2p_lists_its_own_stragglers_when_no_momentum_pass_did(oracle_binding, monkeypatch):
    """g2p skips the particles outside their arena that the momentum pass (or p2g2) of the same
    step listed.  Run on its own -- here a second time on the same float grid, and once right
    after the particles were replaced -- it has no such list and must build one.  (The drift rule is
    off, so that no re-bin takes the stragglers away.)"""
    monkeypatch.setenv("FLOWMPM_DRIFT_DIVISOR", "0")
    grid, n = (40, 24, 16), 6000
    rec = moving_particles(n, grid, seed=53, vel=3.0)
    h = _handle(grid, rec, _lib.ALGO_TILED, rebin_interval=100000)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    for k in range(8):
        for kern in (_lib.KERNEL_CLEAR_GRID, _lib.KERNEL_P2G_MASS, _lib.KERNEL_P2G_MOMENTUM, _lib.KERNEL_UPDATE_GRID,
                     _lib.KERNEL_G2P, _lib.KERNEL_G2P):
            h.run_kernel(kern)
        orc.clear_grid(); orc.p2g1(want, u); orc.p2g2(want, u); orc.update_grid(u)
        orc.g2p(want, u); orc.g2p(want, u)
        assert_bits_equal(h.download_particles(), want, "particles after step %d" % k)
    assert h.timings()["rebin"]["stragglers"] > 0
    # the same particles uploaded again: the list of the last momentum pass is void
    h.run_kernel(_lib.KERNEL_CLEAR_GRID); h.run_kernel(_lib.KERNEL_P2G_MASS); h.run_kernel(_lib.KERNEL_P2G_MOMENTUM)
    h.run_kernel(_lib.KERNEL_UPDATE_GRID)
    orc.clear_grid(); orc.p2g1(want, u); orc.p2g2(want, u); orc.update_grid(u)
    h.upload_particles(want)
    h.run_kernel(_lib.KERNEL_G2P)
    orc.g2p(want, u)
    assert_bits_equal(h.download_particles(), want, "particles after g2p on re-uploaded particles")
    h.close()


def test_tile_bits_survive_a_change_of_particles_between_the_p2g_passes(oracle_binding):
    """clearGrid and updateGrid only visit the tiles the mass pass marked.  When the particle set
    is replaced between the two p2g passes (here: by itself, which still forces a re-bin) the
    marks no longer vouch for the momentum pass, and the device has to fall back to visiting
    every tile -- over several steps, so that a skipped tile would show as stale grid words."""
    grid, n = (40, 24, 24), 6000
    rec = moving_particles(n, grid, seed=31, vel=2.0)
    h = _handle(grid, rec, _lib.ALGO_TILED)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    for k in range(6):
        h.run_kernel(_lib.KERNEL_CLEAR_FIXED_GRID)
        h.run_kernel(_lib.KERNEL_P2G_MASS)
        if k % 2 == 1:
            h.upload_particles(h.download_particles())
        h.run_kernel(_lib.KERNEL_P2G_MOMENTUM)
        h.run_kernel(_lib.KERNEL_UPDATE_GRID)
        h.run_kernel(_lib.KERNEL_G2P)
        orc.run(want, u, 1, 0.0)
        assert np.array_equal(h.download_grid_i32(), orc.cell), "fixed-point grid at step %d" % k
        assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid at step %d" % k)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_clear_of_the_fixed_grid_alone(oracle_binding, algo):
    """clearGrid without the float grid: updateGrid has to zero the float cells the fluid has
    left, also when a step is cleared twice or the two clears alternate."""
    grid, n = (32, 16, 16), 5000
    rec = moving_particles(n, grid, seed=23, vel=2.0)
    h = _handle(grid, rec, algo)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    u = oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    p2g = ((_lib.KERNEL_P2G_MASS, _lib.KERNEL_P2G_MOMENTUM) if algo == _lib.ALGO_TILED
           else (_lib.KERNEL_P2G1, _lib.KERNEL_P2G2))
    emptied, occupied_before = 0, None
    for k in range(12):
        h.run_kernel(_lib.KERNEL_CLEAR_GRID if k == 5 else _lib.KERNEL_CLEAR_FIXED_GRID)
        if k in (3, 6):
            h.run_kernel(_lib.KERNEL_CLEAR_FIXED_GRID)
        for kern in p2g + (_lib.KERNEL_UPDATE_GRID,):
            h.run_kernel(kern)
        if k == 8:
            h.run_kernel(_lib.KERNEL_UPDATE_GRID)
        h.run_kernel(_lib.KERNEL_G2P)
        orc.run(want, u, 1, 0.0)
        assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid at step %d" % k)
        occupied = orc.cellf.reshape(-1, 4)[:, 3] != 0
        if occupied_before is not None:
            emptied += int((occupied_before & ~occupied).sum())
        occupied_before = occupied
    assert emptied > 0, "the case must contain cells that empty out"
    assert_bits_equal(h.download_particles(), want, "particles after 12 steps")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("rebin_interval", [1, 3, 50])
def test_many_steps_match_oracle(oracle_binding, algo, rebin_interval):
    """20 steps: stale bins (rebin_interval > 1) must not change a single bit."""
    if algo == _lib.ALGO_SCATTER and rebin_interval != 1:
        pytest.skip("re-binning is a tiled-path parameter")
    grid, n = (32, 16, 16), 5000
    rec = moving_particles(n, grid, seed=22, vel=1.5)       # fast particles cross tiles quickly
    h = _handle(grid, rec, algo, rebin_interval=rebin_interval)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    h.step(20, 1 / 60)
    orc.run(want, oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS), 20, 1 / 60)
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid")
    assert_bits_equal(h.download_particles(), want, "particles after 20 steps")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 257, 4097])
def test_empty_and_ragged_particle_counts(oracle_binding, algo, n):
    grid = (16, 16, 16)
    rec = moving_particles(n, grid, seed=30 + n) if n else np.zeros((0, 28), np.float32)
    h = _handle(grid, rec, algo)
    h.step(2, 0.5)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    if n:
        orc.run(want, oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS), 2, 0.5)
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid")
    if n:
        assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_particles_on_the_walls_and_non_multiple_of_tile_grid(oracle_binding, algo):
    """Positions on the clamp planes [2, g-2] and a grid that is not a multiple of the tile."""
    grid, n = (21, 13, 10), 3000
    rng = np.random.default_rng(5)
    rec = moving_particles(n, grid, seed=23, vel=2.0)
    lo, hi = 2.0, np.array(grid, np.float32) - 2.0
    rec[:, 0:3] = lo + rng.random((n, 3), dtype=np.float32) * (hi - lo)
    rec[: n // 6, 0] = 2.0
    rec[n // 6: n // 3, 0] = hi[0]
    rec[n // 3: n // 2, 1] = hi[1]
    rec[n // 2: 2 * n // 3, 2] = 2.0
    h = _handle(grid, rec, algo, rebin_interval=2)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    h.step(6, 1 / 60)
    orc.run(want, oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS), 6, 1 / 60)
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_dense_cluster_wraps_like_the_reference(oracle_binding, algo):
    """Thousands of particles in a few cells: the i32 grid overflows and must wrap modulo 2^32
    exactly as atomicAdd<i32> does (SIM:43, 326-329)."""
    grid, n = (16, 16, 16), 9000
    rng = np.random.default_rng(8)
    rec = moving_particles(n, grid, seed=24)
    rec[:, 0:3] = (7.0 + rng.random((n, 3), dtype=np.float32) * 2.0).astype(np.float32)
    h = _handle(grid, rec, algo)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    h.step(1)
    orc.step(want, oracle_binding.make_uniforms(grid, n, **DEFAULT_UNIFORMS))
    assert int(np.abs(orc.cell[:, 3].astype(np.int64)).max()) > 0
    assert n * 1e7 > 2 ** 31 * 8             # far more mass than 8 cells can hold unwrapped
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_mouse_force_and_zero_noise(oracle_binding, algo):
    grid, n = (32, 16, 16), 4000
    uni = dict(DEFAULT_UNIFORMS, noise=0.0, cursor_interaction=1, mouse_ray_direction=(0.0, 0.0, 1.0),
               mouse_ray_origin=(16.0, 8.0, 0.0), mouse_force=(0.8, -0.3, 0.2))
    rec = moving_particles(n, grid, seed=25)
    h = _handle(grid, rec, algo, uniforms=uni)
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    h.step(3, 0.0)
    orc.run(want, oracle_binding.make_uniforms(grid, n, **uni), 3, 0.0)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_noise_with_a_negative_clock(oracle_binding, algo):
    """With positions and clock not negative every argument of the noise's tri() is +0 or a
    positive normal number and the device uses v_fract_f32; a negative clock makes arguments
    negative, where only x - floor(x) is the reference's fract (it rounds to 1.0 for tiny
    negative x, the instruction clamps below 1), so the device must take its general path."""
    grid, n = (32, 16, 16), 4000
    rec = moving_particles(n, grid, seed=26)
    for time in (-0.0, -2.75, -1.0e-30):
        uni = dict(DEFAULT_UNIFORMS, noise=1.5, time=time)
        h = _handle(grid, rec, algo, uniforms=uni)
        orc = oracle_binding.Oracle(grid)
        want = rec.copy()
        h.step(3, 0.0)
        orc.run(want, oracle_binding.make_uniforms(grid, n, **uni), 3, 0.0)
        assert_bits_equal(h.download_particles(), want, "particles, time %r" % time)
        h.close()


def test_six_hundred_steps_of_riding_and_mixing_rebins_leave_the_bits_alone():
    """2^20 particles on 128^3 with the reference's default noise, 600 steps: the device re-bins about every
    fifth step, every re-bin rides on its step's launches, and every fourth one also changes the order of
    the particles inside the tiles (interleaved lanes in the mass pass, runs of ranks changing places, the
    momentum pass and g2p reading with a stride on that step).  None of that may show in the results: the
    reference's own algorithm (one thread per particle, global atomics, no binning at all) must give the same
    particles and the same fixed-point grid, bit for bit."""
    grid, n = (128, 128, 128), 1 << 20
    uni = dict(DEFAULT_UNIFORMS, noise=1.0, rest_density=n / (0.268 * 128 ** 3))
    outs = []
    for algo in (_lib.ALGO_TILED, _lib.ALGO_SCATTER):
        h = flow_amd.Handle(grid, n, algorithm=algo, rebin_interval=64)
        h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=77)
        h.set_uniforms(**uni)
        for _ in range(6):
            h.step(100, 1.0 / 60)
        outs.append((h.download_grid_i32(), h.download_particles(), h.timings()["rebin"].get("performed") or 0))
        h.close()
    assert outs[0][2] >= 40, "the run is meant to hold many re-bins, ten or more of them mixing: %d" % outs[0][2]
    assert np.array_equal(outs[0][0], outs[1][0])
    assert_bits_equal(outs[0][1], outs[1][1], "particles tiled vs scatter after 600 steps")


def test_scatter_and_tiled_agree_at_full_size():
    """Benchmark configuration 2 (2^20 particles, 128^3): too big for the oracle in a test, so
    check what is size-independent: both algorithms produce identical bits, and the fixed-point
    grid mass equals the sum over particles of their 27 truncated weights."""
    grid, n = (128, 128, 128), 1 << 20
    rec = reference_scene(n, grid, seed=0xF10E5EED + 2)
    uni = dict(DEFAULT_UNIFORMS, noise=0.0, rest_density=n / (0.268 * 128 ** 3))
    outs = []
    for algo in (_lib.ALGO_SCATTER, _lib.ALGO_TILED):
        h = _handle(grid, rec, algo, uniforms=uni, rebin_interval=4)
        h.step(6, 0.0)
        outs.append((h.download_grid_i32(), h.download_particles()))
        h.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert_bits_equal(outs[0][1], outs[1][1], "particles scatter vs tiled")
    # mass bookkeeping on the state the grid was built from (positions before the last g2p are
    # gone, so rebuild the grid from the current particles with p2g1 alone)
    h = _handle(grid, outs[1][1], _lib.ALGO_TILED, uniforms=uni)
    h.run_kernel(_lib.KERNEL_CLEAR_GRID)
    h.run_kernel(_lib.KERNEL_P2G1)
    cell = h.download_grid_i32()
    h.close()
    pos = outs[1][1][:, 0:3]
    d = (pos - np.floor(pos)) - np.float32(0.5)
    w = [(np.float32(0.5) * (np.float32(0.5) - d)) * (np.float32(0.5) - d), np.float32(0.75) - d * d,
         (np.float32(0.5) * (np.float32(0.5) + d)) * (np.float32(0.5) + d)]
    want = 0
    for gx in range(3):
        for gy in range(3):
            for gz in range(3):
                wt = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                want += int(np.trunc((wt * np.float32(1e7)).astype(np.float64)).sum())
    assert int(cell[:, 3].astype(np.int64).sum()) == want


def test_permutation_invariance_on_gpu():
    grid, n = (64, 32, 32), 100000
    rec = moving_particles(n, grid, seed=26)
    perm = np.random.default_rng(3).permutation(n)
    a = _handle(grid, rec, _lib.ALGO_TILED)
    b = _handle(grid, np.ascontiguousarray(rec[perm]), _lib.ALGO_TILED)
    a.step(2)
    b.step(2)
    assert np.array_equal(a.download_grid_i32(), b.download_grid_i32())
    assert_bits_equal(a.download_particles()[perm], b.download_particles(), "permuted particles")
    a.close()
    b.close()


def test_dam_break_cluster_matches_oracle(oracle_binding):
    grid, n = (32, 16, 16), 30000          # ~27 particles per cell in the column
    rec = dam_break(n, grid, seed=5)
    uni = dict(DEFAULT_UNIFORMS, noise=0.0, rest_density=20.0)
    h = _handle(grid, rec, _lib.ALGO_TILED, uniforms=uni, rebin_interval=2)
    orc = oracle_binding.Oracle(grid, threads=4)
    want = rec.copy()
    h.step(4, 0.0)
    orc.run(want, oracle_binding.make_uniforms(grid, n, **uni), 4, 0.0)
    assert np.array_equal(h.download_grid_i32(), orc.cell)
    assert_bits_equal(h.download_particles(), want, "dam-break particles")
    h.close()


def test_set_num_particles_and_fields(oracle_binding):
    """conf.particles < conf.maxParticles: only the first ids are simulated (SIM:2513-2522);
    the rest keep their uploaded state."""
    grid, total, live = (32, 16, 16), 3000, 1200
    rec = moving_particles(total, grid, seed=27)
    h = _handle(grid, rec, _lib.ALGO_TILED)
    h.set_num_particles(live)
    h.step(3, 0.1)
    got = h.download_particles()
    orc = oracle_binding.Oracle(grid)
    want = rec.copy()
    orc.run(want, oracle_binding.make_uniforms(grid, live, **DEFAULT_UNIFORMS), 3, 0.1)
    assert_bits_equal(got, want, "all records, live and parked")
    assert_bits_equal(h.download_field(_lib.FIELD_POSITION), want[:live, 0:3], "positions")
    assert_bits_equal(h.download_field(_lib.FIELD_VELOCITY), want[:live, 4:7], "velocities")
    assert_bits_equal(h.download_field(_lib.FIELD_COLOR), want[:live, 24:27], "colors")
    h.set_num_particles(total)             # growing the live set resumes the parked records
    h.step(1)
    orc2 = oracle_binding.Oracle(grid)
    orc2.step(want, oracle_binding.make_uniforms(grid, total, **DEFAULT_UNIFORMS))
    assert_bits_equal(h.download_particles(), want, "after growing the live set")
    h.close()


def test_errors_are_loud():
    grid = (16, 16, 16)
    h = flow_amd.Handle(grid, 10)
    bad = np.zeros((1, 28), np.float32)
    bad[0, 0:3] = (0.5, 8, 8)                      # stencil would leave the grid
    with pytest.raises(_lib.FlowMpmError) as e:
        h.upload_particles(bad)
    assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(_lib.FlowMpmError) as e:
        h.upload_particles(np.zeros((11, 28), np.float32) + 8)
    assert e.value.code == _lib.ERR_CAPACITY
    with pytest.raises(_lib.FlowMpmError) as e:
        h.run_kernel(99)
    assert e.value.code == _lib.ERR_INVALID
    h.close()
    with pytest.raises(_lib.FlowMpmError):
        flow_amd.Handle((4, 4, 4), 10)


def test_simulator_class_follows_the_reference_frame_loop(oracle_binding):
    """MlsMpmSimulator.update() drives the library as the reference's update() drives three.js:
    default scene, conf defaults, 3 frames."""
    from flow_amd.conf import Conf
    sim = flow_amd.MlsMpmSimulator(None, conf=Conf(), rebin_interval=2)
    sim.conf.maxParticles, sim.conf.particles = 16384, 8192
    sim.conf.updateParams()
    sim.init()
    rec = sim.particleBuffer.records().copy()
    for frame in range(3):
        sim.update(1 / 30, 0.5 * frame)            # interval is clamped to 1/60 -> dt = 0.1
    got = sim.readAoS()
    grid, live = (128, 64, 64), 8192
    orc = oracle_binding.Oracle(grid, threads=4)
    want = rec.copy()
    for frame in range(3):
        u = oracle_binding.make_uniforms(grid, live, dt=(1 / 60) * 6, stiffness=3.0, rest_density=0.25,
                                         dynamic_viscosity=0.1, noise=1.0, time=0.5 * frame)
        orc.step(want, u)
    assert_bits_equal(got, want, "simulator records after 3 frames")
    assert_bits_equal(sim.getPositions(), want[:live, 0:3], "getPositions")
    assert_bits_equal(sim.getDensities()[:, 0], want[:live, 3], "getDensities")


def test_default_scene_100_steps_with_state_dumps(oracle_binding):
    """Benchmark configuration 1: the reference's default scene (81,920 live of 262,144 allocated
    particles, 128x64x64, conf defaults, dt = 0.1, clock = step/60), 100 steps, state compared at
    steps 1 and 100 -- grid and every particle record bit for bit."""
    grid, total, live = (128, 64, 64), 262144, 81920
    uni = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.0)
    rec = reference_scene(total, grid, seed=0xF10E5EED)
    h = _handle(grid, rec, _lib.ALGO_TILED, uniforms=uni, rebin_interval=8)
    h.set_num_particles(live)
    orc = oracle_binding.Oracle(grid, threads=8)
    want = rec.copy()
    u = oracle_binding.make_uniforms(grid, live, **uni)
    done = 0
    for checkpoint in (1, 100):
        h.set_uniforms(**dict(uni, time=done / 60))
        h.step(checkpoint - done, 1 / 60)
        u.time = done / 60
        orc.run(want, u, checkpoint - done, 1 / 60)
        done = checkpoint
        assert np.array_equal(h.download_grid_i32(), orc.cell), "fixed-point grid at step %d" % checkpoint
        assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid at step %d" % checkpoint)
        assert_bits_equal(h.download_particles(), want, "particle records at step %d" % checkpoint)
    h.close()


def test_long_run_conservation_tracking():
    """2,000 steps at 2^18 particles: what the step conserves is tracked every 100 steps.
    Fixed-point grid mass stays N*1e7 minus at most 27 truncation units per particle (each of
    the 27 weights loses < 1 unit; they sum to 1 within a few ulp), the particle count and the
    particle masses never change, positions stay inside the clamp box, nothing becomes NaN, and
    the grid momentum of a step equals the sum of its per-cell velocity x mass read-back."""
    grid, n = (64, 64, 64), 1 << 18
    rec = reference_scene(n, grid, seed=77)
    uni = dict(DEFAULT_UNIFORMS, rest_density=n / (0.268 * 64 ** 3))
    h = _handle(grid, rec, _lib.ALGO_TILED, uniforms=uni, rebin_interval=8)
    mass0 = rec[:, 7].copy()
    history = []
    for block in range(20):
        h.set_uniforms(**dict(uni, time=block * 100 / 60))
        h.step(100, 1 / 60)
        cell = h.download_grid_i32()
        p = h.download_particles()
        total_mass = int(cell[:, 3].astype(np.int64).sum())
        history.append(total_mass)
        assert n * 10_000_000 - 40 * n <= total_mass <= n * 10_000_000 + 40 * n
        assert p.shape[0] == n and np.array_equal(p[:, 7], mass0)
        assert np.isfinite(p[:, :23]).all()
        assert (p[:, 0:3] >= 2.0).all() and (p[:, 0:3] <= np.array(grid, np.float32) - 2.0).all()
        # updateGrid stores momentum / mass per live cell; multiplying back reproduces the
        # decoded fixed-point momentum to rounding (sanity link between the two grids)
        f = h.download_grid_f32()
        interior = (f[:, 3] > 0)
        back = f[interior, 0:3].astype(np.float64) * f[interior, 3:4].astype(np.float64)
        fixed = cell[interior, 0:3].astype(np.float64) / 1e7
        zeroed = (f[interior, 0:3] == 0)          # boundary band components are zeroed by updateGrid
        assert np.allclose(np.where(zeroed, fixed, back), fixed, rtol=1e-5, atol=1e-6)
    assert max(history) - min(history) < 40 * n
    h.close()


def test_grid_sums_match_the_downloaded_grid(oracle_binding):
    grid, n = (24, 16, 12), 4000
    rec = moving_particles(n, grid, seed=77, vel=2.0)
    h = _handle(grid, rec, _lib.ALGO_TILED)
    h.step(3, 0.1)
    cell = h.download_grid_i32().astype(np.int64)
    assert np.array_equal(h.grid_sums(), cell.sum(axis=0))
    assert abs(int(h.grid_sums()[3]) - n * 10_000_000) <= 27 * n      # < 1 unit lost per stencil cell
    h.close()


def _truncated_weight_sum(pos):
    """Sum over particles of their 27 truncated fixed-point stencil weights (SIM:307-312, 329), in
    chunks so that the float64 temporaries stay small at 2^24 particles and more."""
    want = 0
    for lo in range(0, pos.shape[0], 1 << 21):
        p = pos[lo:lo + (1 << 21)]
        d = (p - np.floor(p)) - np.float32(0.5)
        w = [(np.float32(0.5) * (np.float32(0.5) - d)) * (np.float32(0.5) - d), np.float32(0.75) - d * d,
             (np.float32(0.5) * (np.float32(0.5) + d)) * (np.float32(0.5) + d)]
        for gx in range(3):
            for gy in range(3):
                wxy = w[gx][:, 0] * w[gy][:, 1]
                for gz in range(3):
                    want += int(np.trunc(((wxy * w[gz][:, 2]) * np.float32(1e7)).astype(np.float64)).sum())
    return want


def _grid_mass_of(grid, particles, uni):
    h = _handle(grid, particles, _lib.ALGO_TILED, uniforms=uni)
    h.run_kernel(_lib.KERNEL_CLEAR_GRID)
    h.run_kernel(_lib.KERNEL_P2G1)
    total = int(h.grid_sums()[3])
    h.close()
    return total


def test_headline_configuration_scatter_and_tiled_agree():
    """The benchmark's headline configuration itself (2^24 particles, 256^3, bench.py's scene, seed
    and uniforms): far beyond what the oracle finishes in a test, so the size-independent checks --
    the reference's algorithm (scatter) and the tiled one leave identical bits in grid and particles;
    the fixed-point grid mass equals the sum of the particles' truncated weights; and a tiled run
    long enough to contain a re-bin the DEVICE decided on (not the interval) still equals the
    scatter run at its end."""
    import bench
    grid, n = (256, 256, 256), 1 << 24
    rec = bench.make_particles("ellipsoid", n, grid, 0xF10E5EED + 3)
    uni = bench.scene_uniforms(n, grid[0] * grid[1] * grid[2], "ellipsoid", 0.0)
    tiled = _handle(grid, rec, _lib.ALGO_TILED, uniforms=uni, rebin_interval=100000)
    scatter = _handle(grid, rec, _lib.ALGO_SCATTER, uniforms=uni)
    del rec
    tiled.step(3, 0.0)
    scatter.step(3, 0.0)
    assert np.array_equal(tiled.download_grid_i32(), scatter.download_grid_i32()), "grid after 3 steps"
    particles = tiled.download_particles()
    assert_bits_equal(particles, scatter.download_particles(), "particles after 3 steps")
    assert _grid_mass_of(grid, particles, uni) == _truncated_weight_sum(particles[:, 0:3])
    del particles
    tiled.reset_timings()
    tiled.step(117, 0.0)
    scatter.step(117, 0.0)
    rebin = tiled.timings()["rebin"]
    assert rebin["performed"] >= 1, "no device-decided re-bin in 117 steps: %r" % (rebin,)
    assert np.array_equal(tiled.download_grid_i32(), scatter.download_grid_i32()), "grid after 120 steps"
    assert_bits_equal(tiled.download_particles(), scatter.download_particles(), "particles after 120 steps")
    tiled.close()
    scatter.close()


def test_dam_break_configuration_with_the_scripted_mouse_ray():
    """Benchmark configuration 5 at its full size (2^25 particles in a column of a 256^3 box, about 9
    per cell) with bench.py's scripted mouse ray changing the uniforms every step: scatter and tiled
    agree bit for bit, the mouse force has acted, and the grid mass is the particles' truncated weights."""
    import bench
    grid, n = (256, 256, 256), 1 << 25
    rec = bench.make_particles("dam_break", n, grid, 0xF10E5EED + 3)
    uni = bench.scene_uniforms(n, grid[0] * grid[1] * grid[2], "dam_break", 0.0)
    outs = []
    for algo in (_lib.ALGO_TILED, _lib.ALGO_SCATTER):
        h = _handle(grid, rec, algo, uniforms=uni, rebin_interval=100000)
        for k in range(10):
            h.set_uniforms(**dict(uni, **bench.scripted_mouse(k, grid)))
            h.step(1, 0.0)
        outs.append((h.download_grid_i32(), h.download_particles()))
        if algo == _lib.ALGO_TILED:
            still = _handle(grid, rec, algo, uniforms=uni, rebin_interval=100000)   # the same run, no mouse
            still.step(10, 0.0)
            assert not np.array_equal(still.download_field(_lib.FIELD_VELOCITY), outs[0][1][:, 4:7])
            still.close()
        h.close()
    del rec
    assert np.array_equal(outs[0][0], outs[1][0]), "dam-break grid"
    assert_bits_equal(outs[0][1], outs[1][1], "dam-break particles")
    particles = outs[0][1]
    del outs
    assert _grid_mass_of(grid, particles, uni) == _truncated_weight_sum(particles[:, 0:3])


def test_a_stage_between_the_momentum_pass_and_g2p_does_not_lose_the_stragglers(monkeypatch):
    """g2p reuses the straggler list of the momentum pass of the same step.  Any tiled stage run in
    between overwrites that one list; g2p must then list its stragglers itself instead of skipping
    them.  No re-bin after the first (drift rule off) makes sure there are stragglers."""
    monkeypatch.setenv("FLOWMPM_DRIFT_DIVISOR", "0")
    grid, n = (40, 24, 16), 6000
    rec = moving_particles(n, grid, seed=77, vel=3.0)
    tiled = _handle(grid, rec, _lib.ALGO_TILED, rebin_interval=100000)
    scatter = _handle(grid, rec, _lib.ALGO_SCATTER)
    tiled.step(40, 0.0)
    scatter.step(40, 0.0)
    before = tiled.timings()["rebin"]["stragglers"]
    assert before > 0
    for k in (_lib.KERNEL_CLEAR_FIXED_GRID, _lib.KERNEL_P2G_MASS, _lib.KERNEL_P2G_MOMENTUM, _lib.KERNEL_UPDATE_GRID,
              # the float grid is complete; these two touch the fixed-point grid and the straggler list only
              _lib.KERNEL_CLEAR_FIXED_GRID, _lib.KERNEL_P2G_MASS,
              _lib.KERNEL_G2P):
        tiled.run_kernel(k)
    scatter.step(1, 0.0)
    assert_bits_equal(tiled.download_particles(), scatter.download_particles(), "particles after the odd order")
    # the same with p2g1 in between
    for k in (_lib.KERNEL_CLEAR_GRID, _lib.KERNEL_P2G_MASS, _lib.KERNEL_P2G_MOMENTUM, _lib.KERNEL_UPDATE_GRID,
              _lib.KERNEL_CLEAR_FIXED_GRID, _lib.KERNEL_P2G1, _lib.KERNEL_G2P):
        tiled.run_kernel(k)
    scatter.step(1, 0.0)
    assert_bits_equal(tiled.download_particles(), scatter.download_particles(), "particles, p2g1 in between")
    assert tiled.timings()["rebin"]["stragglers"] > before, "the two odd steps had no stragglers to lose"
    tiled.close()
    scatter.close()


def test_per_frame_reads_allocate_nothing_after_the_first():
    """getPositions / getDensities / getColors replace the reference renderers' per-frame buffer binding
    (particleRenderer.js:126-133): a thousand frames of reads make no device or pinned allocation
    beyond those of the first frame, and keep returning the right values."""
    grid, n = (32, 16, 16), 5000
    rec = moving_particles(n, grid, seed=32)
    h = _handle(grid, rec, _lib.ALGO_TILED)
    h.step(1, 0.0)
    for field in (_lib.FIELD_POSITION, _lib.FIELD_DENSITY, _lib.FIELD_COLOR):
        h.download_field(field)
    h.download_particles()
    h.download_grid_i32()
    h.grid_sums()
    before = h.allocation_count()
    assert 1 <= before <= 8
    for frame in range(1000):
        h.step(1, 0.0)
        pos = h.download_field(_lib.FIELD_POSITION)
        if frame % 250 == 0:
            h.download_field(_lib.FIELD_DENSITY)
            h.download_field(_lib.FIELD_COLOR)
            assert_bits_equal(pos, h.download_particles()[:, 0:3], "positions at frame %d" % frame)
            h.download_grid_i32()
            h.grid_sums()
    assert h.allocation_count() == before
    h.close()


@pytest.mark.parametrize("sparse", ["0", "120"], ids=["every_tile_an_item", "sparse_tiles_in_the_tail"])
@pytest.mark.parametrize("form", ["0", "1", "2"], ids=["standalone", "own_count", "mass_pass_count"])
def test_every_form_of_the_rebin_gives_the_same_bits(oracle_binding, monkeypatch, form, sparse):
    """The re-bin of flowmpm_step exists in three forms (FLOWMPM_RIDE_REBINS): standalone, riding on the
    step with a counting launch of its own, riding with the count in the mass pass.  Each must leave the
    oracle's bits -- with a re-bin in every step, with the device deciding, with stages run one by one and
    downloads in between, with the live count changed and with particles replaced while a binning exists.
    FLOWMPM_SPARSE_TILE = 120 (the scene holds about 150 particles per tile) also keeps the particles of
    the emptier tiles out of the work items, at the end of the store, in every one of those situations."""
    monkeypatch.setenv("FLOWMPM_RIDE_REBINS", form)
    monkeypatch.setenv("FLOWMPM_SPARSE_TILE", sparse)
    ob = oracle_binding
    grid, n = (40, 24, 32), 9000
    rec = moving_particles(n, grid, seed=61, vel=2.5)
    u = ob.make_uniforms(grid, n, **DEFAULT_UNIFORMS)
    for interval in (1, 3, 100000):
        h = _handle(grid, rec, _lib.ALGO_TILED, rebin_interval=interval)
        orc = ob.Oracle(grid, threads=4)
        want = rec.copy()
        h.step(5, 0.0)
        orc.run(want, u, 5, 0.0)
        assert_bits_equal(h.download_particles(), want, "form %s, interval %d: 5 steps" % (form, interval))
        assert np.array_equal(h.download_grid_i32(), orc.cell)
        # stage by stage (the standalone re-bin), then steps again
        for kern in (_lib.KERNEL_CLEAR_GRID, _lib.KERNEL_P2G1, _lib.KERNEL_P2G2, _lib.KERNEL_UPDATE_GRID, _lib.KERNEL_G2P):
            h.run_kernel(kern)
        orc.step(want, u)
        h.step(3, 0.0)
        orc.run(want, u, 3, 0.0)
        assert_bits_equal(h.download_particles(), want, "form %s, interval %d: stages, then steps" % (form, interval))
        assert_bits_equal(h.download_grid_f32(), orc.cellf, "float grid")
        # fewer live particles (the parked records keep their state), then all of them again
        h.set_num_particles(5000)
        h.step(2, 0.0)
        orc.run(want, ob.make_uniforms(grid, 5000, **DEFAULT_UNIFORMS), 2, 0.0)
        assert_bits_equal(h.download_particles(), want, "form %s, interval %d: 5000 live" % (form, interval))
        h.set_num_particles(n)
        h.step(2, 0.0)
        orc.run(want, u, 2, 0.0)
        assert_bits_equal(h.download_particles(), want, "form %s, interval %d: all live again" % (form, interval))
        # the particles replaced while a binning exists
        h.upload_particles(rec)
        want = rec.copy()
        h.step(4, 0.0)
        orc.run(want, u, 4, 0.0)
        assert_bits_equal(h.download_particles(), want, "form %s, interval %d: after a new upload" % (form, interval))
        assert_bits_equal(h.download_field(_lib.FIELD_COLOR), want[:, 24:27], "colours travel with the records")
        assert_bits_equal(h.download_field(_lib.FIELD_MASS)[:, 0], want[:, 7], "masses travel with the records")
        t = h.timings()["rebin"]
        if interval == 1:
            assert t["performed"] >= 15, t
        if sparse != "0":
            assert t["stragglers"] > 16 * 100, t      # the tail takes the follow-up kernels' path in every step
        h.close()


@pytest.mark.gpu
def test_two_thousand_steps_of_fast_particles_scatter_and_tiled_agree():
    """A soak run of the device-decided re-bins: 200,000 particles that cross a tile every few steps, the
    reference's default noise, 2,000 steps in calls of varying length with reads in between.  The tiled
    algorithm (dozens of re-bins riding on steps, work items of every size, stragglers in every step) must
    end on the bits of the reference's algorithm run as written."""
    grid, n = (96, 64, 80), 200000
    rec = moving_particles(n, grid, seed=404, vel=1.5)
    uni = dict(DEFAULT_UNIFORMS, noise=1.0, rest_density=2.0)
    outs, rebins = [], 0
    for algo in (_lib.ALGO_SCATTER, _lib.ALGO_TILED):
        h = _handle(grid, rec, algo, uniforms=uni, rebin_interval=64)
        done = 0
        for k, chunk in enumerate([1, 7, 192, 300, 500, 1000]):
            h.step(chunk, 0.01)
            h.set_uniforms(**dict(uni, time=(done + chunk) * 0.01))      # the clock goes on across the calls
            done += chunk
            if k % 2 == 0:
                h.download_field(_lib.FIELD_POSITION)
        assert done == 2000
        outs.append((h.download_grid_i32(), h.download_grid_f32(), h.download_particles()))
        if algo == _lib.ALGO_TILED:
            rebins = h.timings()["rebin"]["performed"]
        h.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert_bits_equal(outs[0][1], outs[1][1], "float grid scatter vs tiled")
    assert_bits_equal(outs[0][2], outs[1][2], "particles scatter vs tiled")
    assert np.isfinite(outs[1][2]).all()
    assert rebins >= 20, rebins
