"""Experiment: does a cell-striped particle order inside each tile remove the LDS bank conflicts?
Orders particles on the host as (tile, rank-in-cell, cell) and times a few steps."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flow_amd
from flow_amd import _lib
from bench import make_particles, scene_uniforms

n, grid = 1 << 24, (256, 256, 256)
rec = make_particles("ellipsoid", n, grid, 0xF10E5EED + 3)
ix = np.floor(rec[:, 0:3]).astype(np.int64)
tile = ((ix[:, 0] // 8) * 32 + ix[:, 1] // 8) * 32 + ix[:, 2] // 8
cell = ((ix[:, 0] % 8) * 8 + ix[:, 1] % 8) * 8 + ix[:, 2] % 8
key = tile * 512 + cell
order = np.argsort(key, kind="stable")
ks = key[order]
start = np.r_[0, np.nonzero(np.diff(ks))[0] + 1]
run_id = np.cumsum(np.r_[0, (np.diff(ks) != 0).astype(np.int64)])
rank = np.arange(n) - start[run_id]
for name, perm in (("random", np.arange(n)), ("cell-sorted", order),
                   ("cell-striped", order[np.lexsort((cell[order], rank, tile[order]))])):
    h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=1000)
    h.upload_particles(np.ascontiguousarray(rec[perm]))
    h.set_uniforms(**scene_uniforms(n, 256 ** 3, "ellipsoid"))
    h.step(1)
    h.synchronize()
    h.set_profiling(True); h.reset_timings()
    h.step(3)
    t = h.timings()
    print(name, {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in t.items()}, flush=True)
    h.close()
