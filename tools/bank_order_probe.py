#!/usr/bin/env python3
"""Upper bound of what an LDS-bank-aware order of the particles inside a tile could give the momentum pass:
the benchmark scene uploaded (a) as generated (random inside each tile after the first re-bin, which is stable),
(b) with the particles of every tile dealt out so that 32 consecutive slots hold 32 different values of
(arena index of the stencil base) mod 32 -- every LDS atomic of a half-wave then hits 32 different banks --
and the per-kernel times of the first steps after each upload."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import scene_uniforms
from flow_amd import Handle, _lib

grid = (256, 256, 256)
n = 1 << 24
h = Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=0xF10E5EED + 4)
rec = h.download_particles()
h.close()
c = np.floor(rec[:, 0:3]).astype(np.int64)
tile = ((c[:, 0] // 8) * 32 + c[:, 1] // 8) * 32 + c[:, 2] // 8
b = (c - 1) - (c // 8) * 8 + 2                      # arena coordinates of the stencil base (rim 2)
cls = ((b[:, 0] * 12 + b[:, 1]) * 12 + b[:, 2]) % 32
# k-th particle of its (tile, class)
order = np.lexsort((cls, tile))
t_s, c_s = tile[order], cls[order]
new = np.r_[True, (t_s[1:] != t_s[:-1]) | (c_s[1:] != c_s[:-1])]
first = np.flatnonzero(new)
k = np.arange(n) - np.repeat(first, np.diff(np.r_[first, n]))
dealt = order[np.lexsort((c_s, k, t_s))]            # tile, then round k, then class
rng = np.random.default_rng(3)
shuffled = np.argsort(tile * 4 + 0 * rng.random(n), kind="stable")
for name, perm in (("as generated", np.arange(n)), ("dealt out by bank class", dealt)):
    hh = Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
    hh.upload_particles(rec[perm])
    hh.set_uniforms(**scene_uniforms(n, 256 ** 3, "ellipsoid", 0.0))
    hh.step(1, 0.0)
    hh.synchronize()
    hh.set_profiling(1)
    hh.reset_timings()
    hh.step(3, 0.0)
    hh.synchronize()
    t = hh.timings()
    print("%-26s" % name, {kname: round(v["ms"] / v["launches"], 4) for kname, v in t.items() if isinstance(v, dict) and v.get("launches")})
    hh.close()
