#!/usr/bin/env python3
"""Host time of one SlabSimulation.step() when nothing blocks: rank 0 of a pretended world of two on
one GPU, with a stand-in for torch.distributed whose operations complete at once (the neighbour's
planes and counts read as zero).  With few particles the GPU has next to nothing to do, so the time
per step is what the host needs to enqueue a step -- the part of a multi-GPU step that cannot be
hidden if it exceeds the GPU's own time per step.  Not a correctness run.

    python tools/slab_host_overhead.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from flow_amd import _lib
from flow_amd.particles import reference_scene
from flow_amd.slab import SlabSimulation


class _Done:
    def wait(self):
        return True


class FakeDist:
    """Enough of torch.distributed for SlabSimulation: a world of two in which the other rank sends
    nothing and zeros."""
    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    @staticmethod
    def isend(*a):
        return _Done()

    @staticmethod
    def irecv(*a):
        return _Done()

    def get_backend(self):
        return "nccl"

    def batch_isend_irecv(self, ops):
        for op in ops:
            if op.op is FakeDist.irecv:
                op.tensor.zero_()
        return [_Done()]

    def all_gather_into_tensor(self, out, t, async_op=False):
        out.zero_()
        out[: t.numel()] = t

    def all_reduce(self, t):
        return None


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    grid = (128, 512, 512)
    n = 1 << 14
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rec = reference_scene(n, grid, seed=3)
    rec[:, 0] = np.clip(rec[:, 0], 2.0, 60.0)          # everything on rank 0's planes
    sim = SlabSimulation(grid, [0, 64, 128], 0, 2, max_particles=4 * n, device=0, algorithm=_lib.ALGO_TILED,
                         rebin_interval=64, dist=FakeDist(), torch_device=dev)
    sim.backend.upload(rec, np.arange(n, dtype=np.int32))
    sim.set_uniforms(dt=0.1, stiffness=3.0, rest_density=1.0, dynamic_viscosity=0.1, noise=0.0, time=0.0)
    for _ in range(20):
        sim.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host time per step: %.1f us (enqueue), %.1f us (until the GPU is done); %d particles, planes of 512 x 512"
          % ((t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6, n))


if __name__ == "__main__":
    main()
