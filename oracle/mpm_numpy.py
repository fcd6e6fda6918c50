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
