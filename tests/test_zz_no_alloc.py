"""flowmpm_download_field_static, the entry point that allocates nothing (the values are staged in a
static device array of the library), against flowmpm_download_field: same bits, on two calls in a
row, the second with fewer particles; the size limit; no allocation."""
import numpy as np
import pytest

import flow_amd
from flow_amd import _lib
from helpers import DEFAULT_UNIFORMS, assert_bits_equal, moving_particles

pytestmark = pytest.mark.gpu

FIELDS = (_lib.FIELD_POSITION, _lib.FIELD_VELOCITY, _lib.FIELD_DENSITY, _lib.FIELD_COLOR, _lib.FIELD_DIRECTION,
          _lib.FIELD_MASS)


def test_static_field_read_equals_the_allocating_one_twice_in_a_row():
    grid, n = (32, 16, 16), 6000
    rec = moving_particles(n, grid, seed=51)
    h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=2)
    h.upload_particles(rec)
    h.set_uniforms(**DEFAULT_UNIFORMS)
    h.step(3, 0.0)
    want = {f: h.download_field(f) for f in FIELDS}
    allocations = h.allocation_count()
    for f in FIELDS:                                  # first call: 6000 particles
        got = h.download_field_static(f)
        assert got.shape == want[f].shape
        assert_bits_equal(got, want[f], "field %d, %d particles" % (f, n))
    h.set_num_particles(2500)                         # second call: fewer particles, same static array
    h.step(2, 0.0)
    want = {f: h.download_field(f) for f in FIELDS}
    allocations = max(allocations, h.allocation_count())     # set_num_particles re-uploads through the staging buffer
    for f in FIELDS:
        got = h.download_field_static(f)
        assert got.shape == (2500, _lib.FIELD_WIDTH[f])
        assert_bits_equal(got, want[f], "field %d, 2500 particles" % f)
    assert_bits_equal(h.download_field_static(_lib.FIELD_POSITION), h.download_particles()[:2500, 0:3], "positions")
    assert h.allocation_count() == allocations, "the static read allocated"
    h.close()


def test_static_field_read_rejects_more_than_its_array_holds():
    grid, n = (64, 64, 64), 262144 + 64
    rec = moving_particles(n, grid, seed=52)
    h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED)
    h.upload_particles(rec)
    with pytest.raises(_lib.FlowMpmError) as e:
        h.download_field_static(_lib.FIELD_POSITION)
    assert e.value.code == _lib.ERR_CAPACITY and "262144" in str(e.value)
    h.set_num_particles(262144)                       # exactly the limit works
    assert_bits_equal(h.download_field_static(_lib.FIELD_POSITION), h.download_field(_lib.FIELD_POSITION), "at the limit")
    h.close()


def test_two_handles_share_the_array_one_after_the_other():
    grid = (32, 16, 16)
    a_rec, b_rec = moving_particles(5000, grid, seed=53), moving_particles(3000, grid, seed=54)
    a, b = flow_amd.Handle(grid, 5000), flow_amd.Handle(grid, 3000)
    a.upload_particles(a_rec)
    b.upload_particles(b_rec)
    for _ in range(2):
        assert_bits_equal(a.download_field_static(_lib.FIELD_POSITION), a_rec[:, 0:3], "handle a")
        assert_bits_equal(b.download_field_static(_lib.FIELD_POSITION), b_rec[:, 0:3], "handle b")
    a.close()
    b.close()
