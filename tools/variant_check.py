"""Bit-exactness of a library variant (FLOWMPM_LIBRARY) against the scatter algorithm of the same build:
a few thousand fast particles, several steps, re-binning every second step."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import flow_amd
from flow_amd import _lib
from helpers import DEFAULT_UNIFORMS, moving_particles

grid, n = (72, 40, 56), 60000
rec = moving_particles(n, grid, seed=5, vel=2.0)
outs = []
for algo in (_lib.ALGO_SCATTER, _lib.ALGO_TILED):
    h = flow_amd.Handle(grid, n, algorithm=algo, rebin_interval=2)
    h.upload_particles(rec)
    h.set_uniforms(**DEFAULT_UNIFORMS)
    h.step(7, 0.01)
    outs.append((h.download_grid_i32(), h.download_grid_f32(), h.download_particles()))
    h.close()
ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(*outs))
print("variant %s: %s" % (os.environ.get("FLOWMPM_LIBRARY", "default"), "bit-exact vs scatter" if ok else "MISMATCH"))
sys.exit(0 if ok else 1)
