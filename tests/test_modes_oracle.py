"""CPU tests of the oracle's mode branches (orc_g2p_modes) and of the host-side mode set-up
(flow_amd/modes.py).  The oracle has no second restatement for these branches, so it is
checked against the committed digests and against properties that follow from the
reference's formulas (SIM = src/mls-mpm/mlsMpmSimulator.js)."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

import mode_cases
from flow_amd import modes as host_modes
from flow_amd.particles import palette_rgb
from flow_amd.simulator import Uniform, Vector3, _Uniforms
from helpers import assert_bits_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASE = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.375)
f32 = np.float32


def _run(ob, name, rec, steps=1, base=BASE, only_g2p=False):
    uni, fields = mode_cases.case(name)
    u = ob.make_uniforms(mode_cases.GRID, rec.shape[0], **dict(base, **uni))
    m = mode_cases.fill(ob.ModeUniforms, fields)
    orc = ob.Oracle(mode_cases.GRID)
    for _ in range(steps):
        if only_g2p:
            orc.g2p_modes(rec, u, m)
        else:
            orc.step_modes(rec, u, m)
    return rec, u, m


def test_golden_mode_digests(oracle_binding):
    with open(os.path.join(GOLDEN, "mode_cases.json")) as f:
        gold = json.load(f)
    assert sorted(gold["cases"]) == sorted(mode_cases.ALL_CASES)
    for name, want in gold["cases"].items():
        rec, _, _ = _run(oracle_binding, name, mode_cases.particles(gold["particles"]), gold["steps"],
                         base=gold["uniforms"])
        assert hashlib.sha256(rec.tobytes()).hexdigest() == want["sha256_particles"], name
        assert [int(x) for x in rec[:2].view(np.uint32).ravel()] == want["first_records_bits"], name


def test_no_mode_equals_plain_step(oracle_binding):
    ob = oracle_binding
    rec = mode_cases.particles(3000)
    want = rec.copy()
    got, u, _ = _run(ob, "none", rec, 2)
    orc = ob.Oracle(mode_cases.GRID)
    for _ in range(2):
        orc.step(want, u)
    assert_bits_equal(got, want, "zero mode uniforms vs orc_step")


def test_front_gravity_on_an_empty_grid(oracle_binding):
    """SIM:485-488, 1316, 1365-1367: no centre gravity, no noise; v = (0, 0, -0.1 dt) * 0.98."""
    rec = mode_cases.particles(64)
    pos0 = rec[:, 0:3].copy()
    got, u, _ = _run(oracle_binding, "front_gravity", rec, only_g2p=True)     # float grid all zero
    dt = f32(u.dt)
    vz = (f32(-1.0) * f32(0.1)) * dt * f32(0.98)
    # a particle within 3 cells of the z = 0 wall also gets the wall term; take the others
    far = pos0[:, 2] > 4.0
    assert far.sum() > 40
    assert (got[far, 4] == 0).all() and (got[far, 5] == 0).all()
    assert_bits_equal(got[far, 6], np.full(far.sum(), vz, f32), "front gravity velocity")
    assert not got[:, 8:19].any()                                           # C = 0: nothing gathered


def _sphere_push(pos, center, radius, dt):
    """float32 restatement of SIM:517-528 for one particle."""
    to = (pos - center).astype(f32)
    dist = f32(np.sqrt(f32(f32(to[0] * to[0]) + f32(to[1] * to[1])) + f32(to[2] * to[2])))
    if not dist > radius:
        return np.zeros(3, f32)
    toward = -(to / dist).astype(f32)
    strength = f32(f32(dist - radius) * f32(0.3))
    return ((toward * strength).astype(f32) * dt).astype(f32)


def test_sphere_containment_term(oracle_binding):
    """The difference between 'sphere' and 'none' on an empty grid is exactly the containment
    push added before the noise term; check the sign and size through positions."""
    ob = oracle_binding
    rec = mode_cases.particles(500)
    a, u, _ = _run(ob, "none", rec.copy(), only_g2p=True, base=dict(BASE, noise=0.0))
    b, _, m = _run(ob, "sphere", rec.copy(), only_g2p=True, base=dict(BASE, noise=0.0))
    center, radius, dt = np.array([64, 32, 32], f32), f32(25.0), f32(u.dt)
    dist = np.linalg.norm(rec[:, 0:3] - center, axis=1)
    inside = dist < 24.99
    assert inside.any() and (~inside).any()
    assert_bits_equal(a[inside], b[inside], "particles inside the sphere are untouched")
    far_from_walls = ((rec[:, 0:3] > 5) & (rec[:, 0:3] < np.array(mode_cases.GRID) - 5)).all(axis=1)
    pick = np.flatnonzero(~inside & far_from_walls)[:50]
    for i in pick:
        # v = gravity term, then += push: recompute both in float32
        p = rec[i, 0:3]
        pn = (p / (np.array(mode_cases.GRID, f32) - f32(1)) - f32(0.5)).astype(f32)
        ln = f32(np.sqrt(f32(f32(pn[0] * pn[0]) + f32(pn[1] * pn[1])) + f32(pn[2] * pn[2])))
        v = (f32(0) - ((pn / ln).astype(f32) * f32(0.1)).astype(f32) * dt).astype(f32)
        v = (v + _sphere_push(p, center, radius, dt)).astype(f32)
        assert_bits_equal(b[i, 4:7], v, "particle %d velocity" % i)
    # pushed toward the centre
    moved = b[pick, 0:3] - a[pick, 0:3]
    assert ((moved * (center - rec[pick, 0:3])).sum(axis=1) > 0).all()


def _inside_even_odd(px, py, verts):
    inside = False
    n = len(verts)
    for i in range(n):
        x0, y0 = verts[i]
        x1, y1 = verts[(i + 1) % n]
        if (y0 > py) != (y1 > py) and x0 + (x1 - x0) * (py - y0) / (y1 - y0) > px:
            inside = not inside
    return inside


def test_polygon_inside_test_matches_even_odd_rule(oracle_binding):
    """SIM:550-696: particles inside the V are left alone, those outside move toward it."""
    ob = oracle_binding
    rec = mode_cases.particles(2000)
    a, _, _ = _run(ob, "none", rec.copy(), only_g2p=True)
    b, _, _ = _run(ob, "polygon", rec.copy(), only_g2p=True)
    verts = host_modes.polygon_vertices()
    untouched = (a.view(np.uint32) == b.view(np.uint32)).all(axis=1)
    inside = np.array([_inside_even_odd(float(p[0]), float(p[1]), verts) for p in rec])
    assert 100 < inside.sum() < 1900
    assert (untouched == inside).mean() > 0.999       # float32 vs float64 may differ on an edge
    # 'polygon_empty' (no vertices): nobody is inside, nearest point stays (0, 0) at 9999
    c, _, _ = _run(ob, "polygon_empty", rec.copy(), only_g2p=True)
    assert not (a.view(np.uint32) == c.view(np.uint32)).all(axis=1).any()


def test_segment_and_color_sphere_colour_matching(oracle_binding):
    """Only particles of the segment's colour feel the small sphere (SIM:797-818): two runs
    that differ in the small sphere's position differ exactly on those particles."""
    ob = oracle_binding
    rec = mode_cases.particles(1300)
    rgb = palette_rgb()
    # palette entry of each segment colour, in segment order (SIM:766 ... 1260)
    palette_index = (0, 1, 2, 3, 5, 6, 7, 8, 9)
    for k in (0, 4, 8):
        uni, fields = mode_cases.case("segment_%d" % k)
        u = ob.make_uniforms(mode_cases.GRID, rec.shape[0], **dict(BASE, **uni))
        moved = dict(fields)
        centers = list(fields["segment_sphere_center"])
        centers[k] = (centers[k][0] + 3.0, centers[k][1], centers[k][2])
        moved["segment_sphere_center"] = centers
        out = []
        for f in (fields, moved):
            r = rec.copy()
            ob.Oracle(mode_cases.GRID).g2p_modes(r, u, mode_cases.fill(ob.ModeUniforms, f))
            out.append(r)
        differs = (out[0].view(np.uint32) != out[1].view(np.uint32)).any(axis=1)
        want = np.isclose(rec[:, 24:27], rgb[palette_index[k]], atol=1e-3).all(axis=1)
        if k == 2:
            want |= np.isclose(rec[:, 24:27], rgb[12], atol=1e-3).all(axis=1)
        assert differs.any() and (differs <= want).all(), "segment %d touched other colours" % k
        assert (differs == want).mean() > 0.95


def test_image_grid_attraction_points_at_the_target(oracle_binding):
    """SIM:445-482 with noise strength 0 and no random offset: velocity = (target - pos) * 10 dt."""
    ob = oracle_binding
    n = 400
    rec = mode_cases.particles(n)
    uni, fields = mode_cases.case("image_grid")
    fields = dict(fields, grid_noise_strength=0.0, grid_random_offset=0.0)
    u = ob.make_uniforms(mode_cases.GRID, n, **dict(BASE, **dict(uni, dt=0.01)))
    r = rec.copy()
    ob.Oracle(mode_cases.GRID).g2p_modes(r, u, mode_cases.fill(ob.ModeUniforms, fields))
    i = np.arange(n)
    target = np.stack([4.0 + (i % 160) * 0.75, 2.0 + (i // 160) * 0.75, np.full(n, 32.0)], axis=1)
    far = ((rec[:, 0:3] > 6) & (rec[:, 0:3] < np.array(mode_cases.GRID) - 6)).all(axis=1)
    want = (target - rec[:, 0:3]) * 10 * 0.01
    # the soft walls add to particles that would end within 3 cells of a face; skip those
    ahead = rec[:, 0:3] + want * 0.04
    ok = far & ((ahead > 3) & (ahead < np.array(mode_cases.GRID) - 3)).all(axis=1)
    assert ok.sum() > 200
    np.testing.assert_allclose(r[ok, 4:7], want[ok], rtol=2e-5, atol=2e-5)
    assert not r[:, 8:19].any()                      # no gather in grid mode (SIM:1322)


# ---- host-side set-up (flow_amd/modes.py) --------------------------------------------------

def _mode_host(clock):
    u = _Uniforms()
    u.gridMode, u.frontGravityMode = Uniform(0), Uniform(0)
    return u, host_modes.ModeHost(u, Uniform, Vector3, clock=clock)


def test_toggle_grid_mode_table():
    """SIM:1778-1813."""
    t = [5000.0]
    u, mh = _mode_host(lambda: t[0])
    colors = np.zeros((64, 3), np.float32)
    for mode in range(3, 19):
        if mode == 7:
            continue
        mh.toggle(mode, colors, 40)
        m = mh.pack()
        flags = dict(front=u.frontGravityMode.value, cyl=m.cylinder_mode, sph=m.sphere_containment_mode,
                     poly=m.polygon_containment_mode, cs=m.color_spheres_mode, two=m.two_color_sphere_mode)
        assert u.gridMode.value == 0 and not mh.gridMode
        assert flags["front"] == (1 if mode in (3, 4) else 0)
        assert flags["cyl"] == (1 if mode == 4 else 0)
        assert flags["sph"] == (1 if mode == 5 else 0)
        assert flags["cs"] == (1 if mode == 6 else 0)
        assert flags["poly"] == (1 if mode == 8 else 0)
        assert flags["two"] == (1 if mode == 18 else 0)
        assert list(m.segment_mode) == [1 if mode == 9 + k else 0 for k in range(9)]
        for k in range(9):
            assert m.segment_boost_time[k] == pytest.approx(1.4 if mode == 9 + k else 0.0)
        assert m.color_spheres_boost_time == pytest.approx(1.4 if mode == 6 else 0.0)
    mh.toggle(0, colors, 40)
    rgb = palette_rgb().astype(np.float32)
    assert np.array_equal(colors[:40], rgb[np.arange(40) % 13]) and not colors[40:].any()


def test_segment_geometry_and_boost_clock():
    """SIM:1833-1868 (pink) and 2434-2438."""
    t = [10_000.0]
    u, mh = _mode_host(lambda: t[0])
    colors = np.zeros((8, 3), np.float32)
    mh.toggle(9, colors, 8)
    c = u.pinkSphereCenter.value
    assert (c.x, c.y, c.z) == pytest.approx((64 + 40 * math.cos(math.pi / 6), 32 + 40 * math.sin(math.pi / 6), 32))
    nsc = u.negativeSpaceSphereCenter.value
    assert (nsc.x, nsc.y) == pytest.approx((64 + 20 * math.cos(math.pi / 6), 32 + 20 * math.sin(math.pi / 6)))
    t[0] += 400
    mh.update_boost_clocks()
    assert u.populationSegmentPinkBoostTime.value == pytest.approx(1.0)
    t[0] += 5000
    mh.update_boost_clocks()
    assert u.populationSegmentPinkBoostTime.value == 0
    mh.toggle(10, colors, 8)
    c = u.purpleSphereCenter.value
    assert (c.x, c.y) == pytest.approx((64 + 40 * math.cos(math.radians(135)), 32 + 40 * math.sin(math.radians(135))))
    nsc = u.negativeSpaceSphereCenterPurple.value
    assert (nsc.x, nsc.y) == pytest.approx((64 + 20 * math.cos(math.radians(110)), 32 + 20 * math.sin(math.radians(110))))


def test_polygon_and_color_sphere_setup():
    """SIM:1652-1702, 1704-1774."""
    u, mh = _mode_host(lambda: 1.0)
    colors = np.zeros((8, 3), np.float32)
    mh.toggle(8, colors, 8)
    assert u.polygonVertexCount.value == 7
    assert u.polygonVertices[0].value == pytest.approx([4 + 311.096 / 559 * 120, 2 + 125.894 / 302 * 60])
    assert u.polygonVertices[7].value == [0.0, 0.0]
    mh.toggle(6, colors, 8)
    pos = [u.colorSpherePositions[i].value.toArray() for i in range(13)]
    assert pos[0] == [43.0, 11.0, 32.0] and pos[2] == [85.0, 11.0, 32.0]
    assert pos[3] == [32.5, 32.0, 32.0] and pos[6] == [95.5, 32.0, 32.0]
    assert pos[9] == [85.0, 53.0, 32.0] and pos[10] == [0.0, 0.0, 0.0]
    pal = [u.paletteColors[i].value.toArray() for i in range(13)]
    assert pal[9] == pytest.approx([0x5C / 255, 0xCA / 255, 0xD8 / 255]) and pal[12] == pal[9]
    assert pal[4] == pytest.approx([0x15 / 255, 0x39 / 255, 0x43 / 255])     # duplicates removed


def test_image_grid_setup_and_colours(tmp_path):
    """SIM:1565-1643, 2218-2277."""
    gw, gh, sx, sy, spacing = host_modes.image_grid(300, 200, 81920)
    assert gw * gh >= 81920 and abs(gw / gh - 1.5) < 0.02
    assert spacing == min(120 / gw, 60 / gh)
    assert sx == (128 - gw * spacing) / 2 and sy == (64 - gh * spacing) / 2
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    ppm = tmp_path / "img.ppm"
    ppm.write_bytes(b"P6\n# comment\n30 20\n255\n" + img.tobytes())
    assert np.array_equal(host_modes.read_ppm(str(ppm)), img)
    u, mh = _mode_host(lambda: 1.0)
    colors = np.zeros((700, 3), np.float32)
    mh.toggle(7, colors, 600, image=str(ppm))                # usa: closest palette colour
    assert u.gridMode.value == 1 and mh.gridMode and mh.currentImageMode == 7
    gw, gh = u.gridWidth.value, u.gridHeight.value
    assert gw * gh >= 600
    rgb = palette_rgb()
    # particle 0 samples the bottom-right pixel (both axes flipped, SIM:1625-1626)
    want0 = rgb[np.argmin(np.linalg.norm(rgb - img[19, 29] / 255, axis=1))]
    assert colors[0] == pytest.approx(want0)
    assert not colors[600:].any()
    mh.toggle(1, colors, 600, image=img)                     # lerp mode: on the palette-image segment
    target = mh.gridTargetColors
    near = host_modes.closest_palette(target)
    lo, hi = np.minimum(near, target) - 1e-6, np.maximum(near, target) + 1e-6
    assert ((colors[:600] >= lo) & (colors[:600] <= hi)).all()
    frac_exact = np.isclose(colors[:600], target, atol=1e-7).all(axis=1).mean()
    assert 0.25 < frac_exact < 0.55                          # 40 % keep the image colour
    mh.toggle(3, colors, 600)
    assert mh.currentImageMode == 1                          # SIM:2280-2282: kept through mode 3
    mh.toggle(0, colors, 600)
    assert mh.currentImageMode is None


def test_hidden_particles_are_parked_at_the_x_faces(oracle_binding):
    """SIM:441-442, 1332, 1479-1513 under the int reading of `gridSize.x.mul(0.1)`: the first
    hidePercentage percent of the buffer gather nothing and are pulled toward x = 0 or
    x = gridSize.x, chosen by fract(index * 12.9898) < 0.5."""
    ob = oracle_binding
    n = 1000
    rec = mode_cases.particles(n)
    rec[:, 4:7] = 0
    u = ob.make_uniforms(mode_cases.GRID, n, **dict(BASE, noise=0.0, hide_percentage=30.0))
    m = mode_cases.fill(ob.ModeUniforms, mode_cases.case("none")[1])
    orc = ob.Oracle(mode_cases.GRID)
    orc.cellf[:, 0:3] = 0.25                       # a uniform flow on the grid for the others to gather
    got = rec.copy()
    orc.g2p_modes(got, u, m)
    idx = np.arange(n)
    hidden = (idx.astype(f32) / f32(n)) * f32(100.0) < f32(30.0)
    assert hidden.sum() == 300
    assert not got[hidden, 8:19].any() and got[~hidden, 8:19].any()
    inner = ((rec[:, 0:3] > 6) & (rec[:, 0:3] < np.array(mode_cases.GRID) - 6)).all(axis=1)
    h = idx.astype(f32) * f32(12.9898)
    left = (h - np.floor(h)) < f32(0.5)
    target = np.where(left, 0.0, 128.0)
    sel = hidden & inner
    # centre gravity (|v| = 0.01) plus the parking pull (|x - target| * 0.3 * dt) along x
    pull = (target - rec[:, 0]) * 0.3 * 0.1
    assert np.allclose(got[sel, 4], pull[sel], atol=0.011)
    assert (np.sign(got[sel, 4]) == np.sign(pull[sel])).all()
    assert np.allclose(got[~hidden & inner, 4], 0.25, atol=0.02)   # visible particles follow the grid


@pytest.mark.parametrize("name", mode_cases.ALL_CASES)
def test_numpy_twin_agrees_with_c_on_every_mode_case(oracle_binding, name):
    """Two restatements of SIM:431-1536 that share no code (oracle/mpm_oracle.c and
    oracle/mpm_numpy.py) must agree bit for bit, starting from a grid with flow on it."""
    from oracle import mpm_numpy
    ob = oracle_binding
    n = 3000
    uni, fields = mode_cases.case(name)
    kw = dict(BASE, **uni)
    rec = mode_cases.particles(n, seed=17)
    orc = ob.Oracle(mode_cases.GRID)
    u = ob.make_uniforms(mode_cases.GRID, n, **kw)
    orc.clear_grid()
    orc.p2g1(rec, u)
    orc.p2g2(rec, u)
    orc.update_grid(u)
    want, got = rec.copy(), rec.copy()
    orc.g2p_modes(want, u, mode_cases.fill(ob.ModeUniforms, fields))
    mpm_numpy.g2p_modes(got, orc.cellf, mode_cases.GRID, n, fields, dt=kw["dt"], noise=kw["noise"],
                        time=kw["time"], cursor=kw.get("cursor_interaction", 0),
                        ray_dir=kw.get("mouse_ray_direction", (0, 0, 0)),
                        ray_origin=kw.get("mouse_ray_origin", (0, 0, 0)),
                        mouse_force=kw.get("mouse_force", (0, 0, 0)), grid_mode=kw.get("grid_mode", 0),
                        front_gravity_mode=kw.get("front_gravity_mode", 0),
                        hide_percentage=kw.get("hide_percentage", 0.0))
    assert_bits_equal(got, want, name)
