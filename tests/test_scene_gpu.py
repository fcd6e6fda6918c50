"""The device-side scene generator (flowmpm_generate_particles) against the hosts' generator, the
per-plane histogram, the stencil mass sum, and read-backs larger than the staging buffer."""
import numpy as np
import pytest

from helpers import DEFAULT_UNIFORMS, assert_bits_equal
from flow_amd import Handle, _lib
from flow_amd.particles import dam_break, reference_scene
from flow_amd.slab import plane_starts_from_histogram

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["ellipsoid", "dam_break"])
def test_device_scene_equals_the_host_scene(scene):
    """Every record the device generates -- position (rejection-sampled in binary64, rounded once),
    mass, colour, id, zeros elsewhere -- has the bits of the host generator's record."""
    grid, n, seed = (96, 40, 56), 300007, 0xF10E5EED + 9
    if scene == "ellipsoid":
        want = reference_scene(n, grid, seed=seed)
        kind, fill = _lib.SCENE_ELLIPSOID, 0.25
    else:
        want = dam_break(n, grid, seed=seed, fill=0.3)
        kind, fill = _lib.SCENE_DAM_BREAK, 0.3
    h = Handle(grid, n, algorithm=_lib.ALGO_TILED)
    assert h.generate_particles(kind, n, seed=seed, fill=fill) == n
    assert h.num_particles() == (n, n)
    assert_bits_equal(h.download_particles(), want, "generated records")
    # the generated particles step like uploaded ones
    up = Handle(grid, n, algorithm=_lib.ALGO_TILED)
    up.upload_particles(want)
    for x in (h, up):
        x.set_uniforms(**DEFAULT_UNIFORMS)
        x.step(3, 1.0 / 60)
    assert np.array_equal(h.download_grid_i32(), up.download_grid_i32())
    assert_bits_equal(h.download_particles(), up.download_particles(), "after 3 steps")
    h.close()
    up.close()


def test_slab_handles_keep_their_own_planes_of_the_generated_scene():
    grid, n, seed = (64, 24, 24), 100000, 77
    want = reference_scene(n, grid, seed=seed)
    ix = np.floor(want[:, 0]).astype(np.int64)
    single = Handle(grid, n, algorithm=_lib.ALGO_TILED)
    single.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=seed)
    hist = single.plane_histogram()
    assert np.array_equal(hist, np.bincount(ix, minlength=grid[0]))
    starts = plane_starts_from_histogram(hist, 3)
    seen = np.zeros(n, bool)
    for r in range(3):
        x0, x1 = starts[r], starts[r + 1]
        mine = int(hist[x0:x1].sum())
        h = Handle(grid, mine + 5, algorithm=_lib.ALGO_TILED, slab=(x0, x1 - x0))
        assert h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=seed) == mine
        rec, ids = h.download_particles()
        assert rec.shape[0] == mine and not seen[ids].any()
        seen[ids] = True
        assert_bits_equal(rec, want[ids], "slab %d records" % r)
        assert ((ix[ids] >= x0) & (ix[ids] < x1)).all()
        part = h.plane_histogram()
        assert np.array_equal(part[x0:x1], hist[x0:x1]) and part.sum() == mine
        h.close()
    assert seen.all()
    # a slab too small for its share says so
    h = Handle(grid, 10, algorithm=_lib.ALGO_TILED, slab=(starts[1], starts[2] - starts[1]))
    with pytest.raises(_lib.FlowMpmError) as err:
        h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=seed)
    assert err.value.code == _lib.ERR_CAPACITY
    assert h.num_particles() == (0, 0)
    h.close()
    single.close()


def test_stencil_mass_sum_is_the_grid_mass_of_the_next_step():
    grid, n = (48, 32, 32), 60000
    for algo in (_lib.ALGO_SCATTER, _lib.ALGO_TILED):
        h = Handle(grid, n, algorithm=algo)
        h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=5)
        h.set_uniforms(**DEFAULT_UNIFORMS)
        for _ in range(3):
            expected = h.stencil_mass_sum()
            h.step(1, 1.0 / 60)
            assert int(h.grid_sums()[3]) == expected
            assert expected == int(h.download_grid_i32()[:, 3].astype(np.int64).sum())
        h.close()


def test_generator_argument_checks():
    h = Handle((16, 16, 16), 100, algorithm=_lib.ALGO_TILED)
    for args, code in (((7, 10), _lib.ERR_INVALID), ((_lib.SCENE_ELLIPSOID, 101), _lib.ERR_CAPACITY),
                       ((_lib.SCENE_ELLIPSOID, -1), _lib.ERR_INVALID)):
        with pytest.raises(_lib.FlowMpmError) as err:
            h.generate_particles(*args)
        assert err.value.code == code
    with pytest.raises(_lib.FlowMpmError):
        h.generate_particles(_lib.SCENE_ELLIPSOID, 10, first_id=3)       # single domain: ids from 0
    with pytest.raises(_lib.FlowMpmError):
        h.generate_particles(_lib.SCENE_DAM_BREAK, 10, fill=0.0)
    assert h.generate_particles(_lib.SCENE_ELLIPSOID, 0) == 0
    h.close()


def test_transfers_larger_than_the_staging_buffer():
    """The device staging buffer is capped at 256 MiB: 2^22 + 12345 particles are 450 MiB of records,
    so the upload and every read-back pass through it in several windows, and the handle ends up
    holding no more than the cap."""
    grid, n = (128, 128, 128), (1 << 22) + 12345
    rec = reference_scene(n, grid, seed=3)
    rec[:, 4:7] = np.random.default_rng(0).normal(0, 0.2, (n, 3)).astype(np.float32)
    h = Handle(grid, n, algorithm=_lib.ALGO_TILED)
    h.upload_particles(rec)
    assert_bits_equal(h.download_particles(), rec, "records after a chunked upload and download")
    assert_bits_equal(h.download_field(_lib.FIELD_VELOCITY), rec[:, 4:7], "a field in id order")
    h.set_uniforms(**DEFAULT_UNIFORMS)
    h.step(2, 0.0)            # a re-bin has moved the records: id order is no longer slot order
    g = Handle(grid, n, algorithm=_lib.ALGO_SCATTER)
    g.upload_particles(rec)
    g.set_uniforms(**DEFAULT_UNIFORMS)
    g.step(2, 0.0)
    got = h.download_particles()
    assert_bits_equal(got, g.download_particles(), "records after two steps")
    assert_bits_equal(h.download_field(_lib.FIELD_POSITION), got[:, 0:3], "positions in id order")
    assert np.array_equal(h.download_grid_i32(), g.download_grid_i32())
    h.close()
    g.close()
    # slot-order windows (a slab handle) with ids
    s = Handle(grid, n, algorithm=_lib.ALGO_TILED, slab=(0, grid[0]))
    ids = np.random.default_rng(1).permutation(n).astype(np.int32)
    s.upload_particles(rec, ids)
    back, back_ids = s.download_particles()
    assert np.array_equal(back_ids, ids)
    assert_bits_equal(back, rec, "slab records in slot order")
    s.close()
