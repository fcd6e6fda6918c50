#!/usr/bin/env python3
"""World-of-one RCCL rehearsal on a single GPU: makes every torch.distributed call the slab
path makes (init with device_id, all_reduce, all_gather_into_tensor, all_gather, barrier,
batch_isend_irecv on device tensors -- here to itself) with the "nccl" backend, then runs the
transport's exchange() against itself on the buffers a halo exchange uses.  Catches API and
dtype problems of the RCCL path that the gloo rehearsals cannot; says nothing about xGMI.

    python tools/rccl_selfcheck.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    from flow_amd import _lib
    from flow_amd.slab import SlabSimulation, _Transport, balanced_plane_starts

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    report = {}

    t = torch.arange(8, dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    report["all_reduce_int64"] = bool((t.cpu() == torch.arange(8)).all())

    out = torch.empty(4, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, torch.tensor([3, 1, 4, 1], dtype=torch.int64, device=dev))
    report["all_gather_into_tensor"] = out.cpu().tolist() == [3, 1, 4, 1]

    # the slab step's count exchange: asynchronous, finished and copied to pinned memory on a side stream
    side = torch.cuda.Stream(device=dev)
    out2 = torch.empty(4, dtype=torch.int64, device=dev)
    host = torch.empty(4, dtype=torch.int64).pin_memory()
    work = dist.all_gather_into_tensor(out2, torch.tensor([2, 7, 1, 8], dtype=torch.int64, device=dev), async_op=True)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        work.wait()
        host.copy_(out2, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    done.synchronize()
    report["all_gather_async_side_stream"] = host.tolist() == [2, 7, 1, 8]

    e = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    report["all_reduce_max_f64"] = float(e.item()) == 1.5

    g = [torch.zeros(1, dtype=torch.int64, device=dev)]
    dist.all_gather(g, torch.tensor([7], dtype=torch.int64, device=dev))
    report["all_gather_list"] = int(g[0].item()) == 7
    dist.barrier()

    # point-to-point to itself: int32 halo words and float32 records, as the slab path sends them
    tr = _Transport(dist)
    try:
        a = torch.arange(1 << 16, dtype=torch.int32, device=dev)
        b = torch.zeros_like(a)
        rec = torch.rand((1000, 29), dtype=torch.float32, device=dev)
        rec_in = torch.zeros_like(rec)
        tr.exchange([(a, 0), (rec[100:600], 0)], [(b, 0), (rec_in[100:600], 0)])
        torch.cuda.synchronize()
        report["p2p_self_int32"] = bool((a == b).all().item())
        report["p2p_self_rows"] = bool((rec[100:600] == rec_in[100:600]).all().item())
        # posted, then compute enqueued on the current stream, then completed: what the slab step
        # does with a halo exchange and the interior part of a stage
        c = torch.zeros_like(a)
        posted = tr.post([(a, 0)], [(c, 0)])
        m = torch.rand((2048, 2048), device=dev)
        for _ in range(8):
            m = (m @ m).clamp_(0.0, 1.0)
        tr.complete(posted)
        torch.cuda.synchronize()
        report["p2p_self_overlapped"] = bool((a == c).all().item()) and bool(torch.isfinite(m).all().item())
    except Exception as exc:                       # self send/recv is optional in the library
        report["p2p_self"] = "not supported here: %s" % (str(exc).splitlines()[0],)

    # the slab simulation itself over the nccl group (one slab: no neighbour, no migration)
    import numpy as np
    from flow_amd.particles import reference_scene
    grid = (64, 64, 64)
    n = 1 << 16
    recs = reference_scene(n, grid, seed=5)
    starts = balanced_plane_starts(recs[:, 0], grid[0], 1, dist)
    sim = SlabSimulation(grid, starts, 0, 1, max_particles=2 * n, device=0, algorithm=_lib.ALGO_TILED,
                         rebin_interval=8, dist=dist, torch_device=dev)
    sim.scatter_initial_particles(recs, np.arange(n, dtype=np.int32))
    sim.set_uniforms(dt=0.1, stiffness=3.0, rest_density=1.0, dynamic_viscosity=0.1, noise=1.0, time=0.0)
    for _ in range(4):
        sim.step()
    sim.synchronize()
    totals = sim.conservation()
    report["slab_world1_particles"] = int(totals[4]) == n
    sim.close()
    dist.destroy_process_group()
    for k, v in report.items():
        print("%-28s %s" % (k, v))
    bad = [k for k, v in report.items() if v is False]
    if bad:
        raise SystemExit("FAILED: " + ", ".join(bad))
    print("ok")


if __name__ == "__main__":
    main()
