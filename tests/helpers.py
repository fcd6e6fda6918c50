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


# ---- the image modes, restated scalar by scalar from the reference for the tests ----
# (independent of flow_amd/modes.py and host_js/modes.js, which the tests check against these)

PALETTE_HEX = ("#FF56B7", "#BC96F9", "#C5F0F2", "#007487", "#007487", "#153943", "#232356", "#50121A",
               "#FF6425", "#D6F499", "#5CCAD8", "#5CCAD8", "#C5F0F2")      # mlsMpmSimulator.js:2183-2187


def reference_image_grid(width, height, num_particles):
    """mlsMpmSimulator.js:1565-1602 -> the six grid uniforms of an image of width x height pixels."""
    import math
    aspect = width / height
    gh = math.floor(math.sqrt(num_particles / aspect))
    gw = math.floor(gh * aspect)
    while gw * gh < num_particles:
        if gh > 0 and gw / gh < aspect:
            gw += 1
        else:
            gh += 1
    spacing_x, spacing_y = 120 / gw, 60 / gh
    equal = min(spacing_x, spacing_y)
    return dict(grid_width=gw, grid_height=gh, grid_start_x=(128 - gw * equal) / 2,
                grid_start_y=(64 - gh * equal) / 2, grid_spacing_x=equal, grid_spacing_y=equal)


def reference_image_colors(pixels, gw, gh, num_particles):
    """mlsMpmSimulator.js:1605-1632 -> [n, 3] float64 target colours (both axes flipped)."""
    import math
    height, width = pixels.shape[0], pixels.shape[1]
    out = []
    for i in range(num_particles):
        gx, gy = i % gw, i // gw
        if gy >= gh:
            continue
        nx = gx / (gw - 1) if gw > 1 else 0
        ny = gy / (gh - 1) if gh > 1 else 0
        ix = math.floor((1.0 - nx) * (width - 1))
        iy = math.floor((1.0 - ny) * (height - 1))
        out.append([int(pixels[iy, ix, c]) / 255 for c in range(3)])
    return np.array(out, dtype=np.float64)


def reference_closest_palette(color):
    """mlsMpmSimulator.js:2196-2214: the first palette entry at the smallest RGB distance."""
    import math
    best, best_d = None, math.inf
    for hx in PALETTE_HEX:
        p = [int(hx[1:3], 16) / 255, int(hx[3:5], 16) / 255, int(hx[5:7], 16) / 255]
        d = math.sqrt(sum((color[k] - p[k]) ** 2 for k in range(3)))
        if d < best_d:
            best, best_d = p, d
    return best


def sample_image(width=30, height=20, salt=11):
    """A small deterministic RGB image, [height, width, 3] uint8."""
    return ((np.arange(width * height * 3) * 37 + salt) % 256).astype(np.uint8).reshape(height, width, 3)


def write_ppm(path, pixels):
    with open(path, "wb") as f:
        f.write(b"P6\n# made by a test\n%d %d\n255\n" % (pixels.shape[1], pixels.shape[0]) + pixels.tobytes())


def assert_image_mode_colors(got, pixels, mode, num_particles):
    """The renderer-facing colours after toggleGridMode(mode, image): mode 7 writes the closest
    palette colour (mlsMpmSimulator.js:2219-2233); modes 1 and 2 write the image colour or a point
    on the segment between it and its closest palette colour (SIM:2238-2266)."""
    g = reference_image_grid(pixels.shape[1], pixels.shape[0], num_particles)
    target = reference_image_colors(pixels, g["grid_width"], g["grid_height"], num_particles)
    m = min(num_particles, target.shape[0])
    closest = np.array([reference_closest_palette(target[i]) for i in range(m)])
    got = np.asarray(got, dtype=np.float64)[:m]
    if mode == 7:
        assert np.array_equal(got.astype(np.float32), closest.astype(np.float32)), "mode 7 colours"
        return
    span = closest - target
    lerped = 0
    for i in range(m):
        if np.array_equal(got[i].astype(np.float32), target[i].astype(np.float32)):
            continue
        k = int(np.argmax(np.abs(span[i])))
        f = (got[i, k] - target[i, k]) / span[i, k]
        assert -1e-6 <= f <= 1 + 1e-6, "particle %d: lerp factor %r" % (i, f)
        assert np.allclose(got[i], target[i] + f * span[i], atol=1e-6), "particle %d is off the segment" % i
        lerped += 1
    assert 0.45 * m < lerped < 0.75 * m, "%d of %d colours lerped (expected about 60%%)" % (lerped, m)
