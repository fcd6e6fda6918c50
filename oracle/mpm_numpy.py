"""Second, independent restatement of the reference MLS-MPM step in numpy float32
-- TEST INFRASTRUCTURE ONLY (see oracle/mpm_oracle.h for scope and conventions).

It exists to pin the C oracle: both follow the same reference lines
(src/mls-mpm/mlsMpmSimulator.js, "SIM" below; src/common/noise.js) but share no
code, and tests require them to agree bit for bit.  numpy evaluates every
float32 multiply / add / divide / sqrt as a separate IEEE operation, so no
fused multiply-add can appear here.
"""
import numpy as np

F = np.float32
STRIDE = 28
POS, DENSITY, VEL, MASS, C0, DIR, COLOR = 0, 3, 4, 7, 8, 20, 24
FPM = F(1.0e7)  # SIM:43


def _f2i(x):
    """WGSL i32(f32): truncate toward zero, saturate, NaN -> 0."""
    x = np.asarray(x, dtype=np.float32)
    out = np.zeros(x.shape, dtype=np.int64)
    ok = ~np.isnan(x)
    clipped = np.clip(x[ok].astype(np.float64), -2147483648.0, 2147483647.0)
    out[ok] = np.trunc(clipped).astype(np.int64)
    return out.astype(np.int32)


def encode(x):  # SIM:279-281
    return _f2i(np.asarray(x, dtype=F) * FPM)


def decode(i):  # SIM:282-284
    return i.astype(F) / FPM


def _fract(x):
    return x - np.floor(x)


def _weights(pos):
    """SIM:307-312.  Returns cell index [n,3] and w[k] as [3][n,3]."""
    ci = _f2i(pos) - 1
    d = _fract(pos) - F(0.5)
    w0 = (F(0.5) * (F(0.5) - d)) * (F(0.5) - d)
    w1 = F(0.75) - d * d
    w2 = (F(0.5) * (F(0.5) + d)) * (F(0.5) + d)
    return ci, (w0, w1, w2)


def _ptr(cx, cy, cz, grid):  # SIM:286-290
    return cx * (grid[1] * grid[2]) + cy * grid[2] + cz


def _scatter_add(cell, idx, comp, val):
    flat = cell.reshape(-1).view(np.uint32)
    np.add.at(flat, idx.astype(np.int64) * 4 + comp, val.view(np.uint32))


def clear_grid(cell, cellf):  # SIM:262-277
    cell[...] = 0
    cellf[...] = 0


def p2g1(P, cell, grid, n):  # SIM:295-333
    pos, vel = P[:n, POS:POS + 3], P[:n, VEL:VEL + 3]
    Cm = [P[:n, C0 + 4 * c:C0 + 4 * c + 3] for c in range(3)]  # columns
    ci, w = _weights(pos)
    for gx in range(3):
        for gy in range(3):
            for gz in range(3):
                weight = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                cx, cy, cz = ci[:, 0] + gx, ci[:, 1] + gy, ci[:, 2] + gz
                d = np.stack([(cx.astype(F) + F(0.5)) - pos[:, 0],
                              (cy.astype(F) + F(0.5)) - pos[:, 1],
                              (cz.astype(F) + F(0.5)) - pos[:, 2]], axis=1)
                Q = (Cm[0] * d[:, 0:1] + Cm[1] * d[:, 1:2]) + Cm[2] * d[:, 2:3]
                vc = weight[:, None] * (vel + Q)
                idx = _ptr(cx, cy, cz, grid)
                for r in range(3):
                    _scatter_add(cell, idx, r, encode(vc[:, r]))
                _scatter_add(cell, idx, 3, encode(weight))


def p2g2(P, cell, grid, n, dt, stiffness, rest_density, viscosity):  # SIM:336-392
    pos = P[:n, POS:POS + 3]
    ci, w = _weights(pos)
    density = np.zeros(n, dtype=F)
    for gx in range(3):
        for gy in range(3):
            for gz in range(3):
                weight = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                idx = _ptr(ci[:, 0] + gx, ci[:, 1] + gy, ci[:, 2] + gz, grid)
                density = density + decode(cell[idx, 3]) * weight
    P[:n, DENSITY] = P[:n, DENSITY] * (F(1.0) - F(0.05)) + density * F(0.05)
    with np.errstate(all="ignore"):
        volume = F(1.0) / density
        ratio = density / F(rest_density)
        pow5 = ((ratio * ratio) * (ratio * ratio)) * ratio
        pressure = np.fmax(F(0.0), (pow5 - F(1.0)) * F(stiffness))
        eq16 = [[None] * 3 for _ in range(3)]
        for c in range(3):
            for r in range(3):
                s = -pressure if c == r else np.zeros(n, dtype=F)
                strain = P[:n, C0 + 4 * c + r] + P[:n, C0 + 4 * r + c]
                s = s + strain * F(viscosity)
                eq16[c][r] = ((volume * F(-4.0)) * s) * F(dt)
        for gx in range(3):
            for gy in range(3):
                for gz in range(3):
                    weight = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                    cx, cy, cz = ci[:, 0] + gx, ci[:, 1] + gy, ci[:, 2] + gz
                    d = [(cx.astype(F) + F(0.5)) - pos[:, 0],
                         (cy.astype(F) + F(0.5)) - pos[:, 1],
                         (cz.astype(F) + F(0.5)) - pos[:, 2]]
                    idx = _ptr(cx, cy, cz, grid)
                    for r in range(3):
                        m = ((eq16[0][r] * weight) * d[0] + (eq16[1][r] * weight) * d[1]) \
                            + (eq16[2][r] * weight) * d[2]
                        _scatter_add(cell, idx, r, encode(m))


def update_grid(cell, cellf, grid):  # SIM:395-429
    count = grid[0] * grid[1] * grid[2]
    mass = decode(cell[:count, 3])
    live = mass > 0
    with np.errstate(all="ignore"):
        v = [decode(cell[:count, k]) / mass for k in range(3)]
    i = np.arange(count, dtype=np.int32)
    x = i // grid[2] // grid[1]
    y = i // grid[2] % grid[1]
    z = i % grid[2]
    v[0] = np.where((x < 2) | (x > grid[0] - 2), F(0), v[0])
    v[1] = np.where((y < 2) | (y > grid[1] - 2), F(0), v[1])
    v[2] = np.where((z < 2) | (z > grid[2] - 2), F(0), v[2])
    for k in range(3):
        cellf[:count, k] = np.where(live, v[k], cellf[:count, k])
    cellf[:count, 3] = np.where(live, mass, cellf[:count, 3])


def _tri(x):  # noise.js:3-13
    return np.abs(_fract(x) - F(0.5))


def tri_noise3d(position, speed, time):  # noise.js:49-80
    p = [position[:, a].astype(F).copy() for a in range(3)]
    bp = [q.copy() for q in p]
    z = F(1.4)
    rz = [np.zeros_like(p[0]) for _ in range(3)]
    for _ in range(4):
        b2 = [b * F(2.0) for b in bp]
        dg = [_tri(b2[2] + _tri(b2[1] * F(1.0))),
              _tri(b2[2] + _tri(b2[0] * F(1.0))),
              _tri(b2[1] + _tri(b2[0] * F(1.0)))]
        s = F(time) * (F(0.1) * F(speed))
        p = [p[a] + (dg[a] + s) for a in range(3)]
        bp = [b * F(1.8) for b in bp]
        z = z * F(1.5)
        p = [q * F(1.2) for q in p]
        inner = [_tri(p[1]), _tri(p[2]), _tri(p[0])]
        mid = [_tri(p[0] + inner[0]), _tri(p[1] + inner[1]), _tri(p[2] + inner[2])]
        t = [_tri(p[2] + mid[0]), _tri(p[0] + mid[1]), _tri(p[1] + mid[2])]
        rz = [rz[a] + t[a] / z for a in range(3)]
        bp = [b + F(0.14) for b in bp]
    return np.stack(rz, axis=1)


def _normalize(v):
    with np.errstate(all="ignore"):
        ln = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
        return v / ln[:, None]


def g2p(P, cellf, grid, n, dt, noise, time, cursor=0, ray_dir=None, ray_origin=None,
        mouse_force=None):  # SIM:431-443, 484-494, 1316-1362, 1515-1533
    dt = F(dt)
    pos = P[:n, POS:POS + 3].copy()
    gsm1 = np.array([grid[0] - 1, grid[1] - 1, grid[2] - 1], dtype=F)
    v = np.zeros((n, 3), dtype=F)
    pn = _normalize(pos / gsm1 - F(0.5))
    v = v - (pn * F(0.1)) * dt
    nz = tri_noise3d(pos * F(0.015), time, F(0.11)) - F(0.285)
    nz = _normalize(nz) * F(0.28)
    v = v - (nz * F(noise)) * dt

    ci, w = _weights(pos)
    B = [[np.zeros(n, dtype=F) for _ in range(3)] for _ in range(3)]
    for gx in range(3):
        for gy in range(3):
            for gz in range(3):
                weight = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                cx, cy, cz = ci[:, 0] + gx, ci[:, 1] + gy, ci[:, 2] + gz
                d = [(cx.astype(F) + F(0.5)) - pos[:, 0],
                     (cy.astype(F) + F(0.5)) - pos[:, 1],
                     (cz.astype(F) + F(0.5)) - pos[:, 2]]
                idx = _ptr(cx, cy, cz, grid)
                wv = cellf[idx, 0:3] * weight[:, None]
                for c in range(3):
                    for r in range(3):
                        B[c][r] = B[c][r] + wv[:, r] * d[c]
                v = v + wv
    if cursor == 1:
        rd = np.asarray(ray_dir, dtype=F)
        ro = np.asarray(ray_origin, dtype=F)
        mf = np.asarray(mouse_force, dtype=F)
        b = [pos[:, 0] * F(1.0) - ro[0], pos[:, 1] * F(1.0) - ro[1], pos[:, 2] * F(0.4) - ro[2]]
        cr = [rd[1] * b[2] - rd[2] * b[1], rd[2] * b[0] - rd[0] * b[2], rd[0] * b[1] - rd[1] * b[0]]
        dist = np.sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2])
        f = np.fmax(F(1.0) - dist * F(0.1), F(0.0))
        force = f * f
        v = v + (mf * F(1.0))[None, :] * force[:, None]
        v = v * P[:n, MASS][:, None]
    for c in range(3):
        for r in range(3):
            P[:n, C0 + 4 * c + r] = B[c][r] * F(4.0)
    pos = pos + v * dt
    hi = np.array([grid[0] - 2, grid[1] - 2, grid[2] - 2], dtype=F)
    pos = np.minimum(np.maximum(pos, F(2.0)), hi)
    xn = pos + (v * dt) * F(3.0)
    wall_max = np.array(grid, dtype=F) - F(3.0)
    v = np.where(xn < F(3.0), v + (F(3.0) - xn) * F(0.3), v)
    v = np.where(xn > wall_max, v + (wall_max - xn) * F(0.3), v)
    P[:n, POS:POS + 3] = pos
    P[:n, VEL:VEL + 3] = v
    P[:n, DIR:DIR + 3] = P[:n, DIR:DIR + 3] * (F(1.0) - F(0.1)) + v * F(0.1)


def step(P, cell, cellf, grid, n, dt=0.1, stiffness=3.0, rest_density=2.5, viscosity=0.1,
         noise=1.0, time=0.0, **mouse):
    clear_grid(cell, cellf)
    p2g1(P, cell, grid, n)
    p2g2(P, cell, grid, n, dt, stiffness, rest_density, viscosity)
    update_grid(cell, cellf, grid)
    g2p(P, cellf, grid, n, dt, noise, time, **mouse)


# ---- g2p with the mode branches (SIM:431-1536) -------------------------------------------

# SIM:766, 847, 906, 965, 1024, 1083, 1142, 1201, 1260
SEGMENT_COLORS = np.array([[1.0, 0.3372549, 0.717647], [0.737, 0.588, 0.976], [0.773, 0.941, 0.949],
                           [0.0, 0.455, 0.529], [0.082, 0.224, 0.263], [0.137, 0.137, 0.337],
                           [0.314, 0.071, 0.102], [1.0, 0.392, 0.145], [0.839, 0.957, 0.600]], dtype=F)


def _length(v):
    return np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])


def _eased(boost_time, boost_duration):  # SIM:751-753
    bp = np.minimum(np.maximum(F(boost_time) / F(boost_duration), F(0.0)), F(1.0))
    return (bp * bp) * (F(3.0) - bp * F(2.0))


def _push(v, active, direction, strength, dt):
    """v += (direction * strength) * dt on the rows where `active`."""
    with np.errstate(all="ignore"):
        add = (direction * strength[:, None]) * dt
    return np.where(active[:, None], v + add, v)


def _contain(v, active, pos, center, radius, dt, eased=None):
    """Outside the sphere -> pushed back in (SIM:517-528; with `eased`, SIM:780-794)."""
    to = pos - np.asarray(center, dtype=F)[None, :]
    dist = _length(to)
    with np.errstate(all="ignore"):
        toward = -(to / dist[:, None])
    strength = (dist - F(radius)) * F(0.3)
    if eased is not None:
        strength = strength * (F(1.0) * (F(1.0) - eased) + F(20.0) * eased)
    return _push(v, active & (dist > F(radius)), toward, strength, dt)


def _repel(v, active, pos, center, radius, dt, eased):
    """Inside the negative-space sphere -> pushed out (SIM:821-840)."""
    to = pos - np.asarray(center, dtype=F)[None, :]
    dist = _length(to)
    with np.errstate(all="ignore"):
        away = to / dist[:, None]
    strength = ((F(radius) - dist) * F(0.3)) * (F(1.0) * (F(1.0) - eased) + F(20.0) * eased)
    return _push(v, active & (dist < F(radius)), away, strength, dt)


def g2p_modes(P, cellf, grid, n, m, dt=0.1, noise=1.0, time=0.0, cursor=0, ray_dir=(0, 0, 0),
              ray_origin=(0, 0, 0), mouse_force=(0, 0, 0), grid_mode=0, front_gravity_mode=0,
              hide_percentage=0.0, ids=None):
    """`m`: dict of mode uniforms under the field names of orc_mode_uniforms (missing = 0)."""
    dt = F(dt)
    get = lambda key, default=0: m.get(key, default)  # noqa: E731
    pos = P[:n, POS:POS + 3].copy()
    color = P[:n, COLOR:COLOR + 3].copy()
    index = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    percent = (index.astype(np.uint32).astype(F) / F(np.uint32(n))) * F(100.0)        # SIM:441-442
    hide = (F(hide_percentage) > F(0.0)) & (percent < F(hide_percentage))
    show = ~hide
    v = np.zeros((n, 3), dtype=F)

    # SIM:444-494: grid attraction, else front gravity, else centre gravity
    use_grid = show & (grid_mode == 1)
    use_front = ~use_grid & show & (front_gravity_mode == 1)
    use_centre = ~use_grid & ~use_front
    if use_grid.any():
        gw = int(get("grid_width"))
        i32 = index.astype(np.int32)
        gx = (i32 % gw) if gw else np.zeros(n, np.int32)
        gy = (i32 // gw) if gw else i32
        h = (index.astype(F) * F(12.9898) + gx.astype(F) * F(78.233)) + gy.astype(F) * F(37.719)
        rx = _fract(h) * F(2.0) - F(1.0)
        ry = _fract(h * F(43758.5453)) * F(2.0) - F(1.0)
        sx, sy = F(get("grid_spacing_x")), F(get("grid_spacing_y"))
        ox = (rx * sx) * F(get("grid_random_offset"))
        oy = (ry * sy) * F(get("grid_random_offset"))
        target = np.stack([(F(get("grid_start_x")) + gx.astype(F) * sx) + ox,
                           (F(get("grid_start_y")) + gy.astype(F) * sy) + oy,
                           np.full(n, F(get("grid_z")), dtype=F)], axis=1)
        strength = F(get("grid_noise_strength"))
        target = target + (tri_noise3d(pos * F(0.92), F(1.15), time) * strength) * F(2.0)
        zn = tri_noise3d(pos * F(0.85) + np.array([100.0, 200.0, 300.0], dtype=F), F(0.02), time)
        target[:, 2] = target[:, 2] + (zn[:, 2] * strength) * F(2.0)
        to = target - pos
        dist = _length(to)
        with np.errstate(all="ignore"):
            v = _push(v, use_grid, to / dist[:, None], dist * F(10.0), dt)
    front = np.array([0.0, 0.0, -1.0], dtype=F)
    v = np.where(use_front[:, None], v + ((front * F(0.1)) * dt)[None, :], v)
    gsm1 = np.array([grid[0] - 1, grid[1] - 1, grid[2] - 1], dtype=F)
    pn = _normalize(pos / gsm1 - F(0.5))
    v = np.where(use_centre[:, None], v - (pn * F(0.1)) * dt, v)

    if get("cylinder_mode") == 1:                                                       # SIM:497-511
        flat = np.stack([pos[:, 0] - F(get("cylinder_center_x")), pos[:, 1] - F(get("cylinder_center_y")),
                         pos[:, 2] - pos[:, 2]], axis=1)
        flat[:, 2] = F(0.0)
        dist = _length(flat)
        radius = F(get("cylinder_radius"))
        with np.errstate(all="ignore"):
            v = _push(v, show & (dist < radius), flat / dist[:, None], (F(1.0) - dist / radius) * F(0.5), dt)
    if get("sphere_containment_mode") == 1:                                             # SIM:514-529
        v = _contain(v, show, pos, get("sphere_center"), get("sphere_radius"), dt)
    if get("two_color_sphere_mode") == 1:                                               # SIM:532-547
        v = _contain(v, show, pos, get("sphere_center"), get("sphere_radius"), dt)

    if get("polygon_containment_mode") == 1:                                            # SIM:550-696
        verts = np.zeros((16, 2), dtype=F)
        given = np.asarray(get("polygon_vertices", np.zeros((0, 2))), dtype=F).reshape(-1, 2)
        verts[:given.shape[0]] = given
        count = int(get("polygon_vertex_count"))
        px, py = pos[:, 0], pos[:, 1]
        inside = np.zeros(n, dtype=np.int32)
        pick = lambda i: verts[min(i, 15)]  # noqa: E731
        for i in range(count):
            j = 0 if i == count - 1 else i + 1
            v0, v1 = pick(i), pick(j)
            crosses = (v0[1] > py) != (v1[1] > py)
            with np.errstate(all="ignore"):
                t = (py - v0[1]) / (v1[1] - v0[1])
                xi = v0[0] + (v1[0] - v0[0]) * t
            inside = np.where(crosses & (xi > px), (inside + 1) % 2, inside)
        min_dist = np.full(n, F(9999.0), dtype=F)
        nearest = np.zeros((n, 2), dtype=F)
        for i in range(count):
            j = 0 if i == count - 1 else i + 1
            v0, v1 = pick(i), pick(j)
            edge = v1 - v0
            edge_len = np.sqrt(edge[0] * edge[0] + edge[1] * edge[1])
            len_sq = edge_len * edge_len
            tpx, tpy = px - v0[0], py - v0[1]
            if len_sq < F(0.0001):
                t = np.zeros(n, dtype=F)
            else:
                t = np.minimum(np.maximum((tpx * edge[0] + tpy * edge[1]) / len_sq, F(0.0)), F(1.0))
            qx, qy = v0[0] + edge[0] * t, v0[1] + edge[1] * t
            ddx, ddy = px - qx, py - qy
            dist = np.sqrt(ddx * ddx + ddy * ddy)
            closer = dist < min_dist
            min_dist = np.where(closer, dist, min_dist)
            nearest = np.where(closer[:, None], np.stack([qx, qy], axis=1), nearest)
        to = np.stack([nearest[:, 0] - px, nearest[:, 1] - py, pos[:, 2] - pos[:, 2]], axis=1)
        dist = _length(to)
        with np.errstate(all="ignore"):
            v = _push(v, show & (inside == 0), to / dist[:, None], dist * F(0.3), dt)

    if get("color_spheres_mode") == 1:                                                  # SIM:699-759
        palette = np.asarray(get("palette_colors"), dtype=F).reshape(-1, 3)
        centers = np.asarray(get("color_sphere_positions"), dtype=F).reshape(-1, 3)
        min_dist = np.full(n, F(999.0), dtype=F)
        closest = np.zeros(n, dtype=np.int64)
        for i in range(10):
            dist = _length(color - palette[min(i, 9)][None, :])
            closer = dist < min_dist
            min_dist = np.where(closer, dist, min_dist)
            closest = np.where(closer, i, closest)
        eased = _eased(get("color_spheres_boost_time"), get("color_spheres_boost_duration"))
        target = centers[np.minimum(closest, 9)]
        to = pos - target
        dist = _length(to)
        radius = F(get("color_sphere_radius"))
        with np.errstate(all="ignore"):
            toward = -(to / dist[:, None])
        strength = ((dist - radius) * F(0.3)) * (F(1.0) * (F(1.0) - eased) + F(20.0) * eased)
        v = _push(v, show & (dist > radius), toward, strength, dt)

    seg_mode = list(get("segment_mode", [0] * 9))
    for k in range(9):                                                                  # SIM:762-1313
        if seg_mode[k] != 1:
            continue
        is_segment = _length(color - SEGMENT_COLORS[k][None, :]) < F(0.1)
        eased = _eased(get("segment_boost_time")[k], get("segment_boost_duration")[k])
        v = _contain(v, show, pos, get("sphere_center"), get("segment_main_radius")[k], dt, eased)
        v = _contain(v, show & is_segment, pos, get("segment_sphere_center")[k], get("segment_sphere_radius")[k],
                     dt, eased)
        v = _repel(v, show, pos, get("segment_negative_center")[k], get("negative_space_sphere_radius"), dt, eased)

    if grid_mode == 0 and front_gravity_mode == 0:                                      # SIM:1316-1320
        nz = tri_noise3d(pos * F(0.015), time, F(0.11)) - F(0.285)
        nz = _normalize(nz) * F(0.28)
        v = v - (nz * F(noise)) * dt

    B = [[np.zeros(n, dtype=F) for _ in range(3)] for _ in range(3)]
    gather = show & (grid_mode == 0)                                                    # SIM:1332
    if gather.any():
        ci, w = _weights(pos)
        for gx in range(3):
            for gy in range(3):
                for gz in range(3):
                    weight = (w[gx][:, 0] * w[gy][:, 1]) * w[gz][:, 2]
                    cx, cy, cz = ci[:, 0] + gx, ci[:, 1] + gy, ci[:, 2] + gz
                    d = [(cx.astype(F) + F(0.5)) - pos[:, 0], (cy.astype(F) + F(0.5)) - pos[:, 1],
                         (cz.astype(F) + F(0.5)) - pos[:, 2]]
                    wv = cellf[_ptr(cx, cy, cz, grid), 0:3] * weight[:, None]
                    for c in range(3):
                        for r in range(3):
                            B[c][r] = np.where(gather, B[c][r] + wv[:, r] * d[c], B[c][r])
                    v = np.where(gather[:, None], v + wv, v)

    if grid_mode == 0 and cursor == 1:                                                  # SIM:1355-1360
        rd, ro, mf = (np.asarray(a, dtype=F) for a in (ray_dir, ray_origin, mouse_force))
        b = [pos[:, 0] * F(1.0) - ro[0], pos[:, 1] * F(1.0) - ro[1], pos[:, 2] * F(0.4) - ro[2]]
        cr = [rd[1] * b[2] - rd[2] * b[1], rd[2] * b[0] - rd[0] * b[2], rd[0] * b[1] - rd[1] * b[0]]
        dist = np.sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2])
        f = np.fmax(F(1.0) - dist * F(0.1), F(0.0))
        moved = (v + (mf * F(1.0))[None, :] * (f * f)[:, None]) * P[:n, MASS][:, None]
        v = np.where(show[:, None], moved, v)

    for c in range(3):
        for r in range(3):
            P[:n, C0 + 4 * c + r] = B[c][r] * F(4.0)                                    # SIM:1362

    if front_gravity_mode == 1:                                                         # SIM:1365-1368
        v = v * F(0.98)
    if get("color_spheres_mode") == 1:                                                  # SIM:1370-1382
        e = _eased(get("color_spheres_boost_time"), get("color_spheres_boost_duration"))
        v = np.where(show[:, None], v * (F(0.98) * (F(1.0) - e) + F(0.7) * e), v)
    for k in range(9):                                                                  # SIM:1384-1476
        if seg_mode[k] == 1:
            e = _eased(get("segment_boost_time")[k], get("segment_boost_duration")[k])
            v = np.where(show[:, None], v * (F(0.98) * (F(1.0) - e) + F(0.7) * e), v)

    # SIM:1479-1513 with gridSize.x.mul(0.1) evaluated in the left operand's type (int): width 0
    edge = int(grid[0]) * int(0.1)
    left_max, right_min = edge, int(grid[0]) - edge
    use_left = _fract(index.astype(np.uint32).astype(F) * F(12.9898)) < F(0.5)
    target_x = np.where(use_left, left_max // 2, right_min + edge // 2).astype(F)
    dist = np.abs(pos[:, 0] - target_x)
    toward = np.where(pos[:, 0] < target_x, F(1.0), F(-1.0)).astype(F)
    strength = (dist - F(edge // 2)) * F(0.3)
    parked = v[:, 0] + (toward * strength) * dt
    v[:, 0] = np.where(hide & (dist > F(edge // 2)), parked, v[:, 0])

    pos = pos + v * dt                                                                  # SIM:1515-1533
    hi = np.array([grid[0] - 2, grid[1] - 2, grid[2] - 2], dtype=F)
    pos = np.minimum(np.maximum(pos, F(2.0)), hi)
    xn = pos + (v * dt) * F(3.0)
    wall_max = np.array(grid, dtype=F) - F(3.0)
    v = np.where(xn < F(3.0), v + (F(3.0) - xn) * F(0.3), v)
    v = np.where(xn > wall_max, v + (wall_max - xn) * F(0.3), v)
    P[:n, POS:POS + 3] = pos
    P[:n, VEL:VEL + 3] = v
    P[:n, DIR:DIR + 3] = P[:n, DIR:DIR + 3] * (F(1.0) - F(0.1)) + v * F(0.1)
