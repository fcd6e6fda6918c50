"""Experiment: a wave-level segmented integer reduction in front of the momentum pass's LDS atomics
(library built with -DFLOWMPM_WAVE_REDUCE=1), with the particles ordered by cell inside each tile, against
the default library on the same orders.  Prints the stage times of the first steps after an upload in
that order (no re-bin in between moves the particles).

    FLOWMPM_LIBRARY=$PWD/build_variants/wavered.so python tools/exp_wave_reduce.py
"""
import os
import sys

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
order = np.argsort(tile * 512 + cell, kind="stable")
orders = [("as generated", np.arange(n)), ("sorted by cell", order)]
if len(sys.argv) > 1:
    orders = [o for o in orders if o[0].startswith(sys.argv[1])]
for name, perm in orders:
    h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=1000)
    h.upload_particles(np.ascontiguousarray(rec[perm]))
    h.set_uniforms(**scene_uniforms(n, 256 ** 3, "ellipsoid"))
    h.step(1)
    h.synchronize()
    h.set_profiling(True)
    h.reset_timings()
    h.step(3)
    t = h.timings()
    print("%-16s %-40s" % (name, os.path.basename(os.environ.get("FLOWMPM_LIBRARY", "default library"))),
          {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in t.items() if k in ("p2g1", "p2g2", "g2p")}, flush=True)
    h.close()
