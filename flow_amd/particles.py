"""Initial particle states in the reference's record layout.

`reference_scene` restates the host loop of init() (src/mls-mpm/mlsMpmSimulator.js:89-129)
with a seeded counter-based generator in place of the reference's unseeded Math.random,
so the Python host, the JavaScript host (host_js/particles.js) and the tests produce the
same particles.  `dam_break` is the synthetic clustered workload of the benchmark
configurations; the reference has no such scene.
"""
import numpy as np

from .structured_array import PARTICLE_STRUCT, StructuredArray

# SIM:90-104
PALETTE = ("#FF56B7", "#BC96F9", "#C5F0F2", "#007487", "#007487", "#153943", "#232356",
           "#50121A", "#FF6425", "#D6F499", "#5CCAD8", "#5CCAD8", "#C5F0F2")


def palette_rgb():
    """SIM:107-112: hex -> RGB in [0, 1]."""
    return np.array([[int(h[1:3], 16) / 255, int(h[3:5], 16) / 255, int(h[5:7], 16) / 255]
                     for h in PALETTE], dtype=np.float64)


def _hash32(x):
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def uniform01(seed, index, draw):
    """Draw number `draw` of particle `index`: a double in [0, 1) from a 32-bit hash.
    Counter-based, so any subset of particles can be generated independently."""
    index = np.asarray(index, dtype=np.uint64)
    h = _hash32(index ^ np.uint64(seed & 0xFFFFFFFF))
    h = _hash32((h + np.uint64(draw) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))
    return h.astype(np.float64) / 4294967296.0


def reference_scene(count, grid, seed=0xF10E5EED, first_id=0):
    """Records for particle ids [first_id, first_id + count): position rejection-sampled
    in the unit ball and mapped to ((u*0.8 + 1)/2) * gridSize, mass = 1 - rand*0.002,
    colour = palette[id % 13]; velocity, C, density, direction zero (SIM:114-129)."""
    ids = np.arange(first_id, first_id + count, dtype=np.uint64)
    u = np.empty((count, 3), dtype=np.float64)
    pending = np.arange(count)
    attempt = 0
    while pending.size:
        cand = np.stack([uniform01(seed, ids[pending], 3 * attempt + k) for k in range(3)], axis=1)
        cand = cand * 2.0 - 1.0
        ok = np.sqrt((cand * cand).sum(axis=1)) <= 1.0      # reference loops while dist > 1
        u[pending[ok]] = cand[ok]
        pending = pending[~ok]
        attempt += 1
    pos = (u * 0.8 + 1.0) / 2.0 * np.asarray(grid, dtype=np.float64)
    mass = 1.0 - uniform01(seed ^ 0x5BD1E995, ids, 0) * 0.002
    return _records(count, pos, mass, ids)


def dam_break(count, grid, seed=0xF10E5EED, first_id=0, fill=0.25):
    """Particles uniform in the column x in [3, fill*gx], y in [3, gy-3], z in [3, gz-3]."""
    ids = np.arange(first_id, first_id + count, dtype=np.uint64)
    g = np.asarray(grid, dtype=np.float64)
    lo = np.array([3.0, 3.0, 3.0])
    hi = np.array([max(fill * g[0], 4.0), g[1] - 3.0, g[2] - 3.0])
    u = np.stack([uniform01(seed, ids, k) for k in range(3)], axis=1)
    pos = lo + u * (hi - lo)
    mass = 1.0 - uniform01(seed ^ 0x5BD1E995, ids, 0) * 0.002
    return _records(count, pos, mass, ids)


def _records(count, pos, mass, ids):
    sa = StructuredArray(PARTICLE_STRUCT, count)
    rec = sa.records()
    rec[:, 0:3] = pos
    rec[:, 7] = mass
    rec[:, 24:27] = palette_rgb()[(ids % np.uint64(13)).astype(np.int64)]
    return rec
