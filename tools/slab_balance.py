#!/usr/bin/env python3
"""Load balance of an N-slab decomposition, measured one slab at a time on ONE GPU.

Builds the scene bench.py builds for `--gpus N`, cuts it into the same slabs, and runs every slab on
its own through SlabSimulation.step() with a stand-in for torch.distributed whose operations complete
at once (an interior slab is closed on itself: what leaves through one face enters through the other).
The dynamics near the cuts are therefore not the real ones; the amount of work per slab is.  Prints, per slab,
planes, particles, work items and the time per step, and the ratio max / mean that bounds the weak
scaling efficiency of the real run from above.

    python tools/slab_balance.py --world 8 [--starts 0,140,...,512] [--only SLAB]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from bench import make_particles, scene_uniforms
from flow_amd import _lib
from flow_amd.slab import SlabSimulation, balanced_plane_starts


class _Done:
    def wait(self):
        return True


class LoneRankDist:
    """Enough of torch.distributed for one SlabSimulation of a larger world.  An interior slab is
    closed on itself: what it sends to its high neighbour comes back as if from the low one and the
    other way round (halo planes as they are, particle records shifted by the slab's width), so
    mass flows in where it flows out and there is no void behind the cuts.  The first and the
    last slab have one neighbour only; their receives read as zero and their leavers are dropped."""
    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def __init__(self, rank, world, width):
        self.rank, self.world, self.width = rank, world, float(width)
        self.closed = 0 < rank < world - 1

    @staticmethod
    def isend(*a):
        return _Done()

    @staticmethod
    def irecv(*a):
        return _Done()

    def get_backend(self):
        return "nccl"

    def batch_isend_irecv(self, ops):
        sends = {op.peer: op.tensor for op in ops if op.op is LoneRankDist.isend}
        for op in ops:
            if op.op is not LoneRankDist.irecv:
                continue
            # what comes from the low neighbour is what went to the high one
            src = sends.get(2 * self.rank - op.peer) if self.closed else None
            if src is None or src.shape != op.tensor.shape:
                op.tensor.zero_()
                continue
            op.tensor.copy_(src)
            if op.tensor.dim() == 2:          # particle records: x is word 0
                op.tensor[:, 0] += self.width if op.peer > self.rank else -self.width
        return [_Done()]

    def all_gather_into_tensor(self, out, t, async_op=False):
        out.zero_()
        m = out.view(self.world, -1)
        m[self.rank] = t
        if self.closed:
            m[self.rank - 1, self.rank] = t[self.rank + 1]
            m[self.rank + 1, self.rank] = t[self.rank - 1]

    def all_reduce(self, t):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--particles-log2", type=int, default=24, help="per slab")
    ap.add_argument("--slab-x-per-gpu", type=int, default=64)
    ap.add_argument("--slab-yz", type=int, default=512)
    ap.add_argument("--scene", default="ellipsoid")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--only", type=int, default=-1, help="run this slab alone (for a profiler)")
    ap.add_argument("--starts", default="", help="comma-separated plane starts instead of the balanced ones")
    args = ap.parse_args()

    world, n_per = args.world, 1 << args.particles_log2
    grid = (args.slab_x_per_gpu * world, args.slab_yz, args.slab_yz)
    cells = grid[0] * grid[1] * grid[2]
    n_total = n_per * world
    chunks = [make_particles(args.scene, n_per, grid, 0xF10E5EED + 4, first_id=r * n_per) for r in range(world)]
    x_all = np.concatenate([c[:, 0] for c in chunks])
    if args.starts:
        starts = [int(s) for s in args.starts.split(",")]
    else:
        starts = balanced_plane_starts(x_all, grid[0], world)
    assert len(starts) == world + 1 and starts[0] == 0 and starts[-1] == grid[0]
    print("grid %s, %d particles, plane starts %s" % (grid, n_total, starts), flush=True)
    owner = np.clip(np.searchsorted(np.asarray(starts), np.floor(x_all).astype(np.int64), side="right") - 1, 0, world - 1)
    del x_all
    uniforms = scene_uniforms(n_total, cells, args.scene, 0.0)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    rows = []
    for r in (range(world) if args.only < 0 else [args.only]):
        rec = np.concatenate([c[owner[k * n_per:(k + 1) * n_per] == r] for k, c in enumerate(chunks)])
        ids = np.concatenate([np.nonzero(owner[k * n_per:(k + 1) * n_per] == r)[0] + k * n_per
                              for k in range(world)]).astype(np.int32)
        sim = SlabSimulation(grid, starts, r, world, max_particles=int(rec.shape[0] * 1.5) + 1024, device=0,
                             algorithm=_lib.ALGO_TILED, rebin_interval=64, dist=LoneRankDist(r, world, starts[r + 1] - starts[r]),
                             torch_device=dev)
        sim.backend.set_domain_particles(n_total)
        sim.backend.upload(np.ascontiguousarray(rec), ids)
        sim.set_uniforms(**uniforms)
        for _ in range(args.warmup):
            sim.step()
        torch.cuda.synchronize()
        h = sim.backend.handle
        h.set_profiling(1)
        h.reset_timings()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sim.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        t = h.timings()
        kern = {k: v["ms"] / v["launches"] for k, v in t.items() if isinstance(v, dict) and v.get("launches")}
        rows.append((r, starts[r + 1] - starts[r], rec.shape[0], ms, kern))
        print("slab %d: %3d planes %9d particles  %.3f ms/step  %d re-bins, %.0f stragglers and %.0f particles sent per step  %s"
              % (r, rows[-1][1], rec.shape[0], ms, t["rebin"]["performed"], t["rebin"]["stragglers"] / args.steps,
                 sim.migrated_out / (args.steps + args.warmup),
                 {k: round(v, 3) for k, v in kern.items()}), flush=True)
        sim._pending_counts = None
        sim.close()
        del sim, rec, ids
        torch.cuda.empty_cache()
    times = np.array([x[3] for x in rows])
    print("time per step: max %.3f ms, mean %.3f ms, max / mean %.3f  (weak scaling efficiency <= %.1f %%)"
          % (times.max(), times.mean(), times.max() / times.mean(), 100.0 * times.mean() / times.max()))


if __name__ == "__main__":
    main()
