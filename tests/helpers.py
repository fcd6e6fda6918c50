"""Shared test inputs: seeded particle states that exercise every term of the step."""
import numpy as np

from flow_amd.particles import reference_scene

DEFAULT_UNIFORMS = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0,
                        time=0.375)


def moving_particles(n, grid, seed=1, vel=0.3, shear=0.05):
    """Reference-scene positions with non-zero velocity, affine C, density and direction."""
    rec = reference_scene(n, grid, seed=seed)
    rng = np.random.default_rng(seed)
    rec[:, 4:7] = rng.normal(0, vel, (n, 3)).astype(np.float32)
    for c in range(3):
        rec[:, 8 + 4 * c:8 + 4 * c + 3] = rng.normal(0, shear, (n, 3)).astype(np.float32)
    rec[:, 3] = rng.uniform(0.5, 3.0, n).astype(np.float32)
    rec[:, 20:23] = rng.normal(0, vel, (n, 3)).astype(np.float32)
    return rec


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bits_equal(got, want, what):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, got.shape, want.shape)
    same = bits(got) == bits(want)
    if not same.all():
        bad = np.argwhere(~same)
        first = tuple(bad[0])
        raise AssertionError("%s: %d of %d words differ; first at %s: got %r want %r"
                             % (what, bad.shape[0], same.size, first, got[first], want[first]))
