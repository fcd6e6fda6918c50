"""Benchmark configuration 4 at its full size on one GPU: 2^27 particles in a 512^3 grid, cut into the
eight slabs `bench.py --gpus 8` cuts it into, against the same problem as one single-domain handle.

Nothing here passes through host particle records (15 GB at this size): the scene is generated on the
device (flowmpm_generate_particles), the balanced cut comes from the device-side plane histogram, and
the comparisons are the fixed-point grid plane by plane, the conservation sums, the per-plane particle
counts and the mass bookkeeping."""
import time

import numpy as np
import pytest

from flow_amd import Handle, _lib
from flow_amd.slab import LocalSlabGroup, plane_starts_from_histogram

pytestmark = pytest.mark.gpu


def test_configuration_4_eight_slabs_equal_the_single_domain_at_full_size():
    import bench
    t_start = time.perf_counter()
    grid, n, world, steps = (512, 512, 512), 1 << 27, 8, 3
    seed = 0xF10E5EED + 4                                   # bench.py's multi-GPU scene
    cells = grid[0] * grid[1] * grid[2]
    uni = bench.scene_uniforms(n, cells, "ellipsoid", noise=1.0)     # the reference's default noise: more motion
    plane = grid[1] * grid[2]

    single = Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
    assert single.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=seed) == n
    hist = single.plane_histogram()
    assert int(hist.sum()) == n
    starts = plane_starts_from_histogram(hist, world)
    widths = np.diff(starts)
    assert len(starts) == world + 1 and widths.min() >= 2
    share = np.array([hist[starts[r]:starts[r + 1]].sum() for r in range(world)])
    assert share.max() / share.mean() < 1.05, "the cut should balance the particles: %r" % share

    group = LocalSlabGroup(grid, starts, [int(c * 1.2) + 4096 for c in share], rebin_interval=64)
    for r, h in enumerate(group.handles):
        assert h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=seed) == share[r]
    single.set_uniforms(**uni)
    group.set_uniforms(**uni)

    # mass bookkeeping from the particles' side, before the first step
    mass0 = single.stencil_mass_sum()
    assert mass0 == sum(h.stencil_mass_sum() for h in group.handles)
    assert n * 10 ** 7 - 30 * n < mass0 <= n * 10 ** 7 + 30 * n      # each of 27 truncations loses < 1 unit

    for k in range(steps):
        single.step(1, 0.0)
        group.step()
        if k == 0:
            assert int(single.grid_sums()[3]) == mass0, "grid mass after the first p2g"
    group.finish()
    assert group.migrated > 0, "the case must move particles between slabs"

    sums = single.grid_sums().astype(np.int64)
    totals = group.conservation()
    assert np.array_equal(totals[:4], sums), "conservation sums: slabs %r, single domain %r" % (totals[:4], sums)
    assert int(totals[4]) == n and sum(group.residents()) == n

    # particles per plane after the steps: every particle sits on the slab that owns its plane
    hist_after = single.plane_histogram()
    for r, h in enumerate(group.handles):
        part = h.plane_histogram()
        x0, x1 = starts[r], starts[r + 1]
        assert np.array_equal(part[x0:x1], hist_after[x0:x1]), "particles per plane on slab %d" % r
        assert int(part.sum()) == int(part[x0:x1].sum())

    # the fixed-point grid, plane by plane (the slabs' local grids start one halo plane below x0)
    want = single.download_grid_i32()
    for r, h in enumerate(group.handles):
        x0, x1 = starts[r], starts[r + 1]
        got = h.download_grid_i32()[plane:(x1 - x0 + 1) * plane]
        same = np.array_equal(got, want[x0 * plane:x1 * plane])
        if not same:
            bad = np.nonzero((got != want[x0 * plane:x1 * plane]).any(axis=1))[0]
            raise AssertionError("slab %d: %d cells differ, first in plane %d" % (r, bad.size, x0 + bad[0] // plane))
    del want
    # the float grid of the cells a slab owns follows from the fixed-point grid; spot-check one slab
    r = world // 2
    x0, x1 = starts[r], starts[r + 1]
    got_f = group.handles[r].download_grid_f32()[plane:(x1 - x0 + 1) * plane]
    want_f = single.download_grid_f32()[x0 * plane:x1 * plane]
    assert np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32)), "float grid of slab %d" % r
    group.close()
    single.close()
    print("configuration 4 on one GPU: %.1f s, cut %r, %d particles migrated in %d steps, farthest hop %d"
          % (time.perf_counter() - t_start, starts, group.migrated, steps, group.max_hop))
