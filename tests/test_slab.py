"""The slab-decomposed path.  CPU: the exchange protocol over gloo with the oracle standing in
for the library, world sizes 2 and 3, against the single-domain oracle.  GPU: the device-side
pieces (plane add, leaver extraction, arrival insertion) in one process and a 2-rank rehearsal
on one GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import DEFAULT_UNIFORMS, assert_bits_equal, moving_particles
from flow_amd import Handle, _lib
from flow_amd.slab import balanced_plane_starts, equal_plane_starts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, out, *extra, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "slab_worker.py"), "--out", str(out)] + list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def _reference_run(oracle_binding, grid, n, steps, vel, dt=DEFAULT_UNIFORMS["dt"]):
    rec = moving_particles(n, grid, seed=41, vel=vel)
    orc = oracle_binding.Oracle(grid)
    for k in range(steps):
        orc.step(rec, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k)))
    return rec, orc


def _check_against_single_domain(out, world, rec, orc, grid):
    per = grid[1] * grid[2]
    seen = np.zeros(rec.shape[0], bool)
    moved = 0
    for r in range(world):
        d = np.load(os.path.join(out, "rank%d.npz" % r))
        starts = d["starts"]
        x0, x1 = int(starts[r]), int(starts[r + 1])
        assert not seen[d["ids"]].any(), "a particle is resident on two ranks"
        seen[d["ids"]] = True
        assert_bits_equal(d["rec"], rec[d["ids"]], "particles on rank %d" % r)
        ix = np.floor(d["rec"][:, 0]).astype(int)
        assert ((ix >= x0) & (ix < x1)).all(), "rank %d holds a particle outside its slab" % r
        assert np.array_equal(d["cell"], orc.cell[x0 * per:x1 * per]), "fixed-point grid of rank %d" % r
        want_totals = np.concatenate([orc.cell.astype(np.int64).sum(axis=0), [rec.shape[0]]])
        assert np.array_equal(d["totals"], want_totals), "conservation sums on rank %d" % r
        assert_bits_equal(d["cellf"], orc.cellf[x0 * per:x1 * per], "float grid of rank %d" % r)
        n = rec.shape[0]
        first_owner = np.arange(n)[d["ids"]] * world // n
        moved += int((first_owner != r).sum())
    assert seen.all(), "particles were lost"
    return moved


@pytest.mark.parametrize("world", [2, 3])
def test_slab_protocol_on_cpu_matches_single_domain(oracle_binding, tmp_path, world):
    grid, n, steps, vel = (24, 12, 12), 2500, 12, 3.0
    _launch(world, tmp_path, "--backend", "oracle", "--grid", "24,12,12", "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel))
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel)
    moved = _check_against_single_domain(tmp_path, world, rec, orc, grid)
    assert moved > 0, "the case must exercise migration"


def test_slab_protocol_balanced_boundaries(oracle_binding, tmp_path):
    grid, n, steps, vel = (32, 10, 10), 2000, 6, 2.0
    _launch(2, tmp_path, "--backend", "oracle", "--grid", "32,10,10", "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--balanced")
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel)
    _check_against_single_domain(tmp_path, 2, rec, orc, grid)


def test_eight_slabs_of_uneven_width_on_cpu_match_single_domain(oracle_binding, tmp_path):
    """The 8-way decomposition: eight ranks over gloo, slab boundaries cut at particle quantiles (widths
    that are no multiple of anything), and a time step large enough that particles cross several
    slabs in one step, so leavers go to owners that are not neighbours."""
    grid, n, steps, vel, dt = (61, 10, 10), 4000, 8, 3.0, 1.0
    _launch(8, tmp_path, "--backend", "oracle", "--grid", "61,10,10", "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--dt", str(dt), "--balanced")
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel, dt=dt)
    moved = _check_against_single_domain(tmp_path, 8, rec, orc, grid)
    assert moved > 0
    hops = [int(np.load(os.path.join(tmp_path, "rank%d.npz" % r))["max_hop"]) for r in range(8)]
    assert max(hops) >= 2, "no particle jumped over a whole slab: %r" % hops
    widths = np.diff(np.load(os.path.join(tmp_path, "rank0.npz"))["starts"])
    assert len(set(widths.tolist())) > 2, "the cut should be uneven: %r" % widths


def test_an_overflow_on_one_rank_ends_every_rank(tmp_path):
    """A leaver buffer too small on rank 1 only: every rank sees it in the count matrix and raises the
    capacity error at the same point, instead of rank 1 failing alone while its neighbours wait for
    its records for ever."""
    _launch(3, tmp_path, "--backend", "oracle", "--grid", "24,12,12", "--particles", "2500", "--steps", "12",
            "--vel", "3.0", "--leaver-capacity-of-rank1", "0", timeout=180)
    for r in range(3):
        with open(os.path.join(tmp_path, "rank%d.capacity" % r)) as f:
            assert "rank 1 sends" in f.read()


def _mode_reference_run(oracle_binding, case_name, n, steps, vel):
    import mode_cases
    uni, fields = mode_cases.case(case_name)
    rec = mode_cases.particles(n, seed=41, vel=vel)
    orc = oracle_binding.Oracle(mode_cases.GRID, threads=8)
    m = mode_cases.fill(oracle_binding.ModeUniforms, fields)
    for k in range(steps):
        u = oracle_binding.make_uniforms(mode_cases.GRID, n, **dict(DEFAULT_UNIFORMS, time=0.25 * k, **uni))
        orc.step_modes(rec, u, m)
    return rec, orc


@pytest.mark.parametrize("case_name", ["image_grid", "everything", "hide_60_sphere_mouse"])
def test_slab_protocol_with_mode_forces(oracle_binding, tmp_path, case_name):
    """The mode branches key on the particle's buffer index (grid mode) and colour, both of
    which must survive migration between slabs."""
    n, steps, vel = 3000, 6, 4.0
    _launch(2, tmp_path, "--backend", "oracle", "--grid", "128,64,64", "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--mode-case", case_name)
    rec, orc = _mode_reference_run(oracle_binding, case_name, n, steps, vel)
    moved = _check_against_single_domain(tmp_path, 2, rec, orc, (128, 64, 64))
    assert moved > 0


def test_plane_starts_helpers():
    assert equal_plane_starts(512, 8) == [0, 64, 128, 192, 256, 320, 384, 448, 512]
    x = np.concatenate([np.full(900, 10.5), np.full(100, 50.5)])
    s = balanced_plane_starts(x, 64, 4)
    assert s[0] == 0 and s[-1] == 64 and all(b - a >= 2 for a, b in zip(s, s[1:]))
    x = np.random.default_rng(0).uniform(2, 62, 100000)
    s = balanced_plane_starts(x, 64, 4)
    counts = np.histogram(np.floor(x), bins=s)[0]
    assert counts.max() / counts.mean() < 1.1


@pytest.mark.gpu
@pytest.mark.parametrize("plane_pairs", [False, True], ids=["planes", "packed_pairs"])
@pytest.mark.parametrize("algo", [_lib.ALGO_SCATTER, _lib.ALGO_TILED])
def test_two_slabs_in_one_process_match_single_domain(oracle_binding, algo, plane_pairs):
    """Drives two slab handles on one GPU with a local 'transport' (device copies), which
    checks the halo merges (whole planes, or the packed plane pairs the step uses), leaver
    extraction and arrival insertion against the oracle."""
    import torch
    from flow_amd.slab import SlabSimulation

    grid, n, steps = (24, 16, 12), 4000, 10
    rec = moving_particles(n, grid, seed=41, vel=3.0)
    starts = [0, 11, 24]

    class LocalDist:
        """Stands in for torch.distributed between two in-process 'ranks'."""
        def __init__(self):
            self.box = {}

    sims = [SlabSimulation(grid, starts, r, 2, max_particles=n, device=0, algorithm=algo, rebin_interval=2)
            for r in range(2)]
    dest = (np.floor(rec[:, 0]) >= starts[1]).astype(int)
    for r, s in enumerate(sims):
        rows = np.nonzero(dest == r)[0]
        s.backend.upload(rec[rows], rows.astype(np.int32))

    def exchange(mask):
        a, b = sims[0].backend, sims[1].backend
        if plane_pairs:
            out_a, out_b = a.new_halo_buffer(mask), b.new_halo_buffer(mask)
            a.halo_pack(_lib.SIDE_HIGH, mask, out_a)
            b.halo_pack(_lib.SIDE_LOW, mask, out_b)
            torch.cuda.synchronize()
            a.halo_add(_lib.SIDE_HIGH, mask, out_b)
            b.halo_add(_lib.SIDE_LOW, mask, out_a)
            torch.cuda.synchronize()
            return
        bufs = [x.new_plane_buffer(2) for x in (a, b)]
        a.plane_read(_lib.PLANE_HIGH_OWNED, bufs[0][0]); a.plane_read(_lib.PLANE_HIGH_HALO, bufs[0][1])
        b.plane_read(_lib.PLANE_LOW_HALO, bufs[1][0]); b.plane_read(_lib.PLANE_LOW_OWNED, bufs[1][1])
        torch.cuda.synchronize()
        a.plane_add(_lib.PLANE_HIGH_OWNED, bufs[1][0], mask); a.plane_add(_lib.PLANE_HIGH_HALO, bufs[1][1], mask)
        b.plane_add(_lib.PLANE_LOW_HALO, bufs[0][0], mask); b.plane_add(_lib.PLANE_LOW_OWNED, bufs[0][1], mask)
        torch.cuda.synchronize()

    moved = 0
    for k in range(steps):
        for s in sims:
            s.set_uniforms(**dict(DEFAULT_UNIFORMS, time=0.25 * k))
            s.backend.run_kernel(_lib.KERNEL_CLEAR_GRID)
            s.backend.run_kernel(_lib.KERNEL_P2G1)
        exchange(_lib.COMP_MASS)
        for s in sims:
            s.backend.run_kernel(_lib.KERNEL_P2G2)
        exchange(_lib.COMP_MOMENTUM)
        for s in sims:
            s.backend.run_kernel(_lib.KERNEL_UPDATE_GRID)
            s.backend.run_kernel(_lib.KERNEL_G2P)
        out = [s.backend.extract_leavers(starts) for s in sims]
        torch.cuda.synchronize()
        for r, (counts, packed) in enumerate(out):
            assert counts[r] == 0
            if counts[1 - r]:
                moved += int(counts[1 - r])
                sims[1 - r].backend.insert_arrivals(packed.clone())
        torch.cuda.synchronize()
    assert moved > 0

    want = rec.copy()
    orc = oracle_binding.Oracle(grid)
    for k in range(steps):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, time=0.25 * k)))
    per = grid[1] * grid[2]
    total = 0
    for r, s in enumerate(sims):
        got, ids = s.download()
        total += got.shape[0]
        assert_bits_equal(got, want[ids], "slab %d particles" % r)
        x0, x1 = starts[r], starts[r + 1]
        cell = s.backend.handle.download_grid_i32()[per:(x1 - x0 + 1) * per]
        assert np.array_equal(cell, orc.cell[x0 * per:x1 * per]), "slab %d grid" % r
        s.close()
    assert total == n


@pytest.mark.gpu
@pytest.mark.parametrize("grid,n,steps,vel", [((24, 16, 12), 4000, 8, 3.0),      # thin slabs: every tile is a boundary tile
                                             ((112, 16, 16), 14000, 10, 3.0),   # 56 planes per slab: interior tiles too
                                             ((3 * 40, 16, 16), 12000, 8, 3.0)])
def test_two_rank_rehearsal_on_one_gpu(oracle_binding, tmp_path, grid, n, steps, vel):
    """Two (three for the last case) processes, all on cuda:0, gloo transport with host
    staging: the whole SlabSimulation code path of the multi-GPU benchmark except RCCL itself --
    the mass pass split around the migration, momentum pass and g2p in an interior and a
    boundary launch with the halo exchanges posted in between."""
    world = 3 if grid[0] == 120 else 2
    _launch(world, tmp_path, "--backend", "gpu", "--grid", "%d,%d,%d" % grid, "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel))
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel)
    moved = _check_against_single_domain(tmp_path, world, rec, orc, grid)
    assert moved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case_name", ["image_grid", "everything", "hide_60_sphere_mouse"])
def test_two_rank_rehearsal_with_mode_forces(oracle_binding, tmp_path, case_name):
    n, steps, vel = 20000, 6, 4.0
    _launch(2, tmp_path, "--backend", "gpu", "--grid", "128,64,64", "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--mode-case", case_name)
    rec, orc = _mode_reference_run(oracle_binding, case_name, n, steps, vel)
    moved = _check_against_single_domain(tmp_path, 2, rec, orc, (128, 64, 64))
    assert moved > 0


@pytest.mark.gpu
def test_eight_slabs_in_one_process_match_single_domain(oracle_binding):
    """Eight slab handles of uneven width on one GPU, the tiled algorithm's own stages (mass pass,
    momentum pass, packed halo pairs), leavers handed to whichever slab owns them -- with a time step
    large enough that some jump over whole slabs -- against the single-domain oracle, bit for bit,
    plus the conservation sums."""
    import torch
    from flow_amd.slab import SlabSimulation

    grid, n, steps, dt = (142, 16, 16), 30000, 8, 1.0
    starts = [0, 37, 48, 57, 66, 74, 83, 95, 142]       # a balanced cut's shape: wide ends, thin middle
    world = len(starts) - 1
    rec = moving_particles(n, grid, seed=43, vel=3.0)
    sims = [SlabSimulation(grid, starts, r, world, max_particles=n, device=0, algorithm=_lib.ALGO_TILED,
                           rebin_interval=3) for r in range(world)]
    dest = np.searchsorted(np.asarray(starts), np.floor(rec[:, 0]).astype(int), side="right") - 1
    for r, s in enumerate(sims):
        rows = np.nonzero(dest == r)[0]
        s.backend.upload(rec[rows], rows.astype(np.int32))

    def exchange(mask):
        packed = [(s.backend.new_halo_buffer(mask), s.backend.new_halo_buffer(mask)) for s in sims]
        for r, s in enumerate(sims):
            if r > 0:
                s.backend.halo_pack(_lib.SIDE_LOW, mask, packed[r][0])
            if r < world - 1:
                s.backend.halo_pack(_lib.SIDE_HIGH, mask, packed[r][1])
        torch.cuda.synchronize()
        for r, s in enumerate(sims):
            if r > 0:
                s.backend.halo_add(_lib.SIDE_LOW, mask, packed[r - 1][1])
            if r < world - 1:
                s.backend.halo_add(_lib.SIDE_HIGH, mask, packed[r + 1][0])
        torch.cuda.synchronize()

    moved, far = 0, 0
    for k in range(steps):
        for s in sims:
            s.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k))
            s.backend.run_kernel(_lib.KERNEL_CLEAR_FIXED_GRID)
            s.backend.run_kernel(_lib.KERNEL_P2G_MASS)
        exchange(_lib.COMP_MASS)
        for s in sims:
            s.backend.run_kernel(_lib.KERNEL_P2G_MOMENTUM)
        exchange(_lib.COMP_MOMENTUM)
        if k == steps - 1:
            sums = sum(s.backend.grid_sums().astype(np.int64) for s in sims)
        for s in sims:
            s.backend.run_kernel(_lib.KERNEL_UPDATE_GRID)
            s.backend.run_kernel(_lib.KERNEL_G2P)
        out = [s.backend.extract_leavers(starts) for s in sims]
        torch.cuda.synchronize()
        for r, (counts, packed) in enumerate(out):
            assert counts[r] == 0
            offs = np.concatenate([[0], np.cumsum(counts)])
            for peer in np.nonzero(counts)[0]:
                moved += int(counts[peer])
                far += int(counts[peer]) if abs(int(peer) - r) >= 2 else 0
                sims[peer].backend.insert_arrivals(packed[int(offs[peer]):int(offs[peer + 1])].clone())
        torch.cuda.synchronize()
    assert moved > 0 and far > 0, "the case must send particles past a neighbour (%d moved, %d far)" % (moved, far)

    want = rec.copy()
    orc = oracle_binding.Oracle(grid, threads=8)
    for k in range(steps):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k)))
    assert np.array_equal(sums, orc.cell.astype(np.int64).sum(axis=0)), "conservation sums over the 8 slabs"
    per = grid[1] * grid[2]
    total = 0
    for r, s in enumerate(sims):
        got, ids = s.download()
        total += got.shape[0]
        assert_bits_equal(got, want[ids], "slab %d particles" % r)
        x0, x1 = starts[r], starts[r + 1]
        cell = s.backend.handle.download_grid_i32()[per:(x1 - x0 + 1) * per]
        assert np.array_equal(cell, orc.cell[x0 * per:x1 * per]), "slab %d grid" % r
        s.close()
    assert total == n


@pytest.mark.gpu
def test_eight_rank_rehearsal_on_one_gpu(oracle_binding, tmp_path):
    """Eight processes on cuda:0 over gloo: SlabSimulation.step() exactly as the 8-GPU benchmark runs it
    (split mass pass, interior / boundary launches, posted halo exchanges, migration to any owner),
    uneven slab widths from the balanced cut, particles that jump over whole slabs."""
    grid, n, steps, vel, dt = (150, 16, 16), 24000, 8, 3.0, 1.0
    _launch(8, tmp_path, "--backend", "gpu", "--grid", "%d,%d,%d" % grid, "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--dt", str(dt), "--balanced")
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel, dt=dt)
    moved = _check_against_single_domain(tmp_path, 8, rec, orc, grid)
    assert moved > 0
    hops = [int(np.load(os.path.join(tmp_path, "rank%d.npz" % r))["max_hop"]) for r in range(8)]
    assert max(hops) >= 2, "no particle jumped over a whole slab: %r" % hops


@pytest.mark.gpu
@pytest.mark.parametrize("rebin_interval,ride", [(1, 2), (3, 2), (2, 0)], ids=["ride_every_step", "ride_every_third", "standalone"])
def test_slab_step_calls_with_riding_rebins_match_single_domain(oracle_binding, rebin_interval, ride):
    """The six-call slab step (flowmpm_slab_step_*) on eight slabs in one process: re-bins ride on the
    step while particles leave (holes filled from the tail: tile keys and ranks move with the movers) and
    arrive (counted by the arrivals' mass pass), with a time step large enough that some jump over whole
    slabs -- against the single-domain oracle, bit for bit."""
    from flow_amd.slab import LocalSlabGroup

    grid, n, steps, dt = (142, 16, 24), 40000, 9, 1.0
    starts = [0, 37, 48, 57, 66, 74, 83, 95, 142]
    world = len(starts) - 1
    rec = moving_particles(n, grid, seed=47, vel=3.0)
    group = LocalSlabGroup(grid, starts, n, rebin_interval=rebin_interval)
    dest = np.searchsorted(np.asarray(starts), np.floor(rec[:, 0]).astype(int), side="right") - 1
    for r, b in enumerate(group.backends):
        rows = np.nonzero(dest == r)[0]
        b.upload(rec[rows], rows.astype(np.int32))
        b.handle.set_option(_lib.OPTION_RIDE_REBINS, ride)
    for k in range(steps):
        group.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k))
        group.step()
    group.finish()
    assert group.migrated > 0 and group.max_hop >= 2
    rebins = [h.timings()["rebin"]["performed"] for h in group.handles]
    assert min(rebins) >= (steps - 1) // rebin_interval, rebins

    want = rec.copy()
    orc = oracle_binding.Oracle(grid, threads=8)
    for k in range(steps):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k)))
    totals = group.conservation()
    assert np.array_equal(totals[:4], orc.cell.astype(np.int64).sum(axis=0)) and int(totals[4]) == n
    per = grid[1] * grid[2]
    seen = np.zeros(n, bool)
    for r, b in enumerate(group.backends):
        got, ids = b.download()
        assert not seen[ids].any()
        seen[ids] = True
        assert_bits_equal(got, want[ids], "slab %d particles" % r)
        x0, x1 = starts[r], starts[r + 1]
        assert np.array_equal(b.handle.download_grid_i32()[per:(x1 - x0 + 1) * per], orc.cell[x0 * per:x1 * per]), "slab %d grid" % r
        assert_bits_equal(b.handle.download_grid_f32()[per:(x1 - x0 + 1) * per], orc.cellf[x0 * per:x1 * per], "slab %d float grid" % r)
    assert seen.all()
    group.close()


@pytest.mark.gpu
def test_slab_step_calls_out_of_order_are_refused():
    h = Handle((32, 16, 16), 100, algorithm=_lib.ALGO_TILED, slab=(8, 16))
    with pytest.raises(_lib.FlowMpmError) as err:
        h.slab_step_interior()
    assert err.value.code == _lib.ERR_INVALID and "flowmpm_slab_step_begin comes next" in str(err.value)
    h.slab_step_begin()
    with pytest.raises(_lib.FlowMpmError):
        h.slab_step_grid()
    h.slab_step_begin()          # a refused call starts the sequence over
    h.slab_step_pack_mass(None, None)
    h.slab_step_interior()
    h.slab_step_boundary(None, None, None, None)
    h.slab_step_grid()
    h.slab_step_end(None, None)
    h.close()
    single = Handle((32, 16, 16), 100, algorithm=_lib.ALGO_TILED)
    with pytest.raises(_lib.FlowMpmError):
        single.slab_step_begin()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("starts", [[0, 37, 48, 57, 66, 74, 83, 95, 142], [0, 142], [0, 70, 142]], ids=["eight", "one", "two"])
def test_native_slab_group_matches_single_domain(oracle_binding, starts):
    """flowmpm_slab_group: the whole decomposed step, migration and halo merges inside the library, all
    slabs on the one GPU -- against the single-domain oracle, bit for bit, with read-backs in id order."""
    from flow_amd import SlabGroup

    grid, n, steps, dt = (142, 16, 24), 40000, 9, 1.0
    rec = moving_particles(n, grid, seed=53, vel=3.0)
    group = SlabGroup(grid, starts, n, rebin_interval=2)
    assert not group.uses_rccl()
    group.upload_particles(rec)
    assert group.num_particles()[0] == n
    for k in range(steps):
        group.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k))
        group.step(1, 0.0)
    want = rec.copy()
    orc = oracle_binding.Oracle(grid, threads=8)
    for k in range(steps):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k)))
    assert_bits_equal(group.download_particles(), want, "particles in id order")
    assert_bits_equal(group.download_field(_lib.FIELD_POSITION), want[:, 0:3], "positions in id order")
    assert_bits_equal(group.download_field(_lib.FIELD_DENSITY), want[:, 3:4], "densities in id order")
    totals = group.conservation()
    assert np.array_equal(totals[:4], orc.cell.astype(np.int64).sum(axis=0)) and int(totals[4]) == n
    per = grid[1] * grid[2]
    for r, h in enumerate(group.handles):
        x0, x1 = starts[r], starts[r + 1]
        assert np.array_equal(h.download_grid_i32()[per:(x1 - x0 + 1) * per], orc.cell[x0 * per:x1 * per]), "slab %d grid" % r
    moved, hop = group.migration_stats()
    if len(starts) > 2:
        assert moved > 0 and (hop >= 2 or len(starts) == 3)
    # several steps in one call with the clock advancing, as flowmpm_step does
    group.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=0.1, time=1.0))
    group.step(4, 1.0 / 60)
    for k in range(4):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=0.1, time=np.float32(1.0 + k / 60))))
    assert_bits_equal(group.download_particles(), want, "particles after a multi-step call")
    group.close()


@pytest.mark.gpu
def test_native_slab_group_generates_its_scene_and_refuses_bad_placements():
    from flow_amd import SlabGroup
    grid, n = (64, 24, 24), 50000
    group = SlabGroup(grid, [0, 30, 64], n)
    group.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=9)
    from flow_amd.particles import reference_scene
    assert_bits_equal(group.download_particles(), reference_scene(n, grid, seed=9), "generated scene through the group")
    group.close()
    with pytest.raises(_lib.FlowMpmError):
        SlabGroup(grid, [0, 30, 63], n)                     # does not end at grid_x
    with pytest.raises(_lib.FlowMpmError):
        SlabGroup(grid, [0, 20, 40, 64], n, devices=[0, 0, 1])     # neither one device nor one each


@pytest.mark.gpu
def test_rccl_world_of_one_sends_to_itself():
    from flow_amd.simulator import rccl_selftest
    rccl_selftest(0, 1 << 18)


@pytest.mark.gpu
def test_torch_distributed_path_over_rccl_in_a_world_of_one():
    """Every torch.distributed call the slab simulation makes (init with a device id, the collectives, the
    asynchronous count exchange on a side stream, batch_isend_irecv of halo words and records on device
    buffers -- here to itself), with the "nccl" backend on the one GPU, and four steps of a one-slab
    SlabSimulation over that group.  Says nothing about xGMI; catches what the gloo rehearsals cannot."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selfcheck.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")
    for line in ("all_gather_async_side_stream", "p2p_self_int32", "p2p_self_overlapped", "slab_world1_particles"):
        assert "%-28s True" % line in r.stdout, r.stdout


def test_limited_plane_starts_rule():
    from flow_amd.slab import limited_plane_starts, plane_starts_from_histogram
    hist = np.zeros(64, dtype=np.int64)
    hist[4:20] = 1000                       # everything sits at low x
    even = [0, 16, 32, 48, 64]
    want = plane_starts_from_histogram(hist, 4)
    assert want == [0, 8, 12, 16, 64]
    # an unlimited budget goes all the way, widths permitting
    assert limited_plane_starts(hist, even, [64] * 4, 10 ** 9) == want
    # a budget of 2.5 planes' particles moves every boundary by at most 2 occupied planes (empty ones are free)
    step = limited_plane_starts(hist, even, [64] * 4, 2500)
    moved = [int(hist[min(a, b):max(a, b)].sum()) for a, b in zip(step, even)]
    assert max(moved) <= 2500 and step != even
    # repeated calls converge on the balanced cut
    cur = even
    for _ in range(20):
        cur = limited_plane_starts(hist, cur, [64] * 4, 2500)
    assert cur == want
    # a slab that may not grow leaves the cut alone
    assert limited_plane_starts(hist, even, [16, 16, 16, 16], 10 ** 9) == even


@pytest.mark.gpu
def test_recutting_the_slabs_while_running_keeps_the_bits(oracle_binding):
    """flowmpm_slab_group_rebalance every other step on particles that stream through the domain: slabs
    change width (planes and their particles change hands through the ordinary migration), and grid and
    particles still equal the single-domain oracle's bit for bit."""
    from flow_amd import SlabGroup

    grid, n, steps, dt = (96, 16, 16), 30000, 12, 1.0
    rec = moving_particles(n, grid, seed=61, vel=2.0)
    rec[:, 0] = 4.0 + (rec[:, 0] - rec[:, 0].min()) * 0.35          # a column at low x ...
    rec[:, 4] = np.abs(rec[:, 4]) + 1.5                              # ... streaming toward high x
    group = SlabGroup(grid, [0, 24, 48, 72, 96], n, rebin_interval=3, slab_max_nx=96)
    group.upload_particles(rec)
    cuts = [list(group.plane_starts)]
    for k in range(steps):
        group.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k))
        group.step(1, 0.0)
        # (not after the last step: a re-cut clears the grids, which the comparisons below read)
        if k % 2 == 1 and k < steps - 1 and group.rebalance(0.5):
            cuts.append(list(group.plane_starts))
    assert len(cuts) >= 3, "the cut should have moved several times: %r" % cuts
    _, per = group.num_particles()
    assert per.max() / per.mean() < 1.6, per
    want = rec.copy()
    orc = oracle_binding.Oracle(grid, threads=8)
    for k in range(steps):
        orc.step(want, oracle_binding.make_uniforms(grid, n, **dict(DEFAULT_UNIFORMS, dt=dt, time=0.25 * k)))
    assert_bits_equal(group.download_particles(), want, "particles after %d re-cuts" % (len(cuts) - 1))
    totals = group.conservation()
    assert np.array_equal(totals[:4], orc.cell.astype(np.int64).sum(axis=0)) and int(totals[4]) == n
    plane = grid[1] * grid[2]
    starts = group.plane_starts
    for r, h in enumerate(group.handles):
        x0, x1 = starts[r], starts[r + 1]
        assert h.local_grid() == (x0 - 1, x1 - x0 + 2)
        assert np.array_equal(h.download_grid_i32()[plane:(x1 - x0 + 1) * plane], orc.cell[x0 * plane:x1 * plane]), "slab %d grid" % r
    # a group created without room to grow refuses politely: the cut stays
    fixed = SlabGroup(grid, [0, 24, 48, 72, 96], n)
    fixed.upload_particles(rec)
    fixed.step(1, 0.0)
    assert fixed.rebalance(0.5) is False and fixed.plane_starts == [0, 24, 48, 72, 96]
    fixed.close()
    group.close()


@pytest.mark.gpu
def test_three_rank_rehearsal_with_recuts(oracle_binding, tmp_path):
    """Three processes on cuda:0 over gloo: SlabSimulation.rebalance() after every second step (histogram
    all-reduce, flowmpm_set_slab, migration of the planes that changed hands) -- still the single domain's
    bits, with slabs that ended up at another width than they started."""
    grid, n, steps, vel = (120, 16, 16), 12000, 8, 3.0
    _launch(3, tmp_path, "--backend", "gpu", "--grid", "%d,%d,%d" % grid, "--particles", str(n),
            "--steps", str(steps), "--vel", str(vel), "--recut-every", "2")
    rec, orc = _reference_run(oracle_binding, grid, n, steps, vel)
    _check_against_single_domain(tmp_path, 3, rec, orc, grid)
    d = np.load(os.path.join(tmp_path, "rank0.npz"))
    assert int(d["recuts"]) >= 1 and list(d["starts"]) != equal_plane_starts(grid[0], 3), list(d["starts"])
