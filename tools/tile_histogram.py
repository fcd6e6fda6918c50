"""Histogram of the particles per 8x8x8 tile after a number of steps of the benchmark scene: how much of
the grid the work items cover, and how many of them hold only a few particles."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import flow_amd
from flow_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=420)
ap.add_argument("--noise", type=float, default=1.0)
ap.add_argument("--particles-log2", type=int, default=24)
ap.add_argument("--grid", default="256,256,256")
ap.add_argument("--scene", default="ellipsoid")
a = ap.parse_args()
grid = tuple(int(x) for x in a.grid.split(","))
n = 1 << a.particles_log2
rec = bench.make_particles(a.scene, n, grid, 0xF10E5EED + 4)
h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
h.upload_particles(rec)
h.set_uniforms(**bench.scene_uniforms(n, grid[0] * grid[1] * grid[2], a.scene, a.noise))
for at in (0, a.steps):
    if at:
        h.step(at, 0.1)
    p = h.download_particles()
    t = (np.floor(p[:, 0]).astype(np.int64) // 8 * (grid[1] // 8) + np.floor(p[:, 1]).astype(np.int64) // 8) * (grid[2] // 8) \
        + np.floor(p[:, 2]).astype(np.int64) // 8
    c = np.bincount(t, minlength=(grid[0] // 8) * (grid[1] // 8) * (grid[2] // 8))
    occ = c[c > 0]
    edges = [1, 8, 32, 128, 512, 1024, 2048, 4096, 1 << 30]
    print("after %d steps: %d of %d tiles occupied, mean %.0f, median %.0f particles" % (at, occ.size, c.size, occ.mean(), np.median(occ)))
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (occ >= lo) & (occ < hi)
        print("  [%5d, %5s): %6d tiles, %5.1f %% of the particles" % (lo, hi if hi < (1 << 30) else "inf", m.sum(), 100.0 * occ[m].sum() / n))
