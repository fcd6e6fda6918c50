#!/usr/bin/env python3
"""What a slab step costs, measured on ONE GPU with every slab of the decomposition present.

Builds the problem `bench.py --gpus N` builds (scene generated on the device, balanced cut), creates all
N slab handles in this process (flow_amd.slab.LocalSlabGroup: the stage order of the multi-GPU step with
device buffers handed from slab to slab instead of a transport) and steps them in lock-step.  The GPU
runs the slabs one after the other, so `ms per slab step` = time per group step / N is what one rank of
the real run computes per step, with the real flows of particles and halo planes through the cuts and
without any transport cost.  With --single the same problem also runs as one single-domain handle, for
the per-particle cost next to it.

    python tools/slab_group_bench.py --world 8 [--steps 40] [--single] [--noise 1.0]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from bench import scene_uniforms
from flow_amd import Handle, _lib
from flow_amd.slab import LocalSlabGroup, plane_starts_from_histogram


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--particles-log2", type=int, default=24, help="per slab")
    ap.add_argument("--slab-x-per-gpu", type=int, default=64)
    ap.add_argument("--slab-yz", type=int, default=512)
    ap.add_argument("--scene", choices=("ellipsoid", "dam_break"), default="ellipsoid")
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--windows", type=int, default=1, help="timed windows of --steps steps, back to back")
    ap.add_argument("--single", action="store_true", help="also time the same problem as one single-domain handle")
    ap.add_argument("--profile-slab", type=int, default=-1, help="HIP-event timing of this slab's launches")
    ap.add_argument("--native", action="store_true",
                    help="drive the slabs through flowmpm_slab_group (the library's own lock-step) instead of "
                         "the Python lock-step over the six step calls")
    ap.add_argument("--recut-every", type=int, default=0,
                    help="--native: flowmpm_slab_group_rebalance after every k-th step (0: the cut stays)")
    ap.add_argument("--recut-fraction", type=float, default=0.1)
    ap.add_argument("--report-every", type=int, default=0, help="--native: resident particles per slab every k steps")
    args = ap.parse_args()

    world, n_per = args.world, 1 << args.particles_log2
    n = n_per * world
    grid = (args.slab_x_per_gpu * world, args.slab_yz, args.slab_yz)
    cells = grid[0] * grid[1] * grid[2]
    seed = 0xF10E5EED + 4
    kind = _lib.SCENE_ELLIPSOID if args.scene == "ellipsoid" else _lib.SCENE_DAM_BREAK
    uni = scene_uniforms(n, cells, args.scene, args.noise)
    hist = _lib.scene_plane_histogram(kind, n, grid, seed=seed)
    starts = plane_starts_from_histogram(hist, world)
    share = [int(hist[starts[r]:starts[r + 1]].sum()) for r in range(world)]
    out = {"grid": list(grid), "particles": n, "world": world, "plane_starts": starts, "particles_per_slab": share,
           "noise": args.noise, "scene": args.scene}
    if args.native:
        from flow_amd import SlabGroup
        cap = int(max(share) * 2.5) + 4096
        group = SlabGroup(grid, starts, cap, rebin_interval=64, slab_max_nx=grid[0] if args.recut_every else 0)
        group.generate_particles(kind, n, seed=seed)
        group.set_uniforms(**uni)
        group.step(args.warmup, 0.0)
        group.synchronize()
        out["windows"], out["balance"] = [], []
        done = 0
        for _ in range(args.windows):
            t0 = time.perf_counter()
            for k in range(args.steps):
                group.step(1, 0.0)
                done += 1
                if args.recut_every and done % args.recut_every == 0:
                    group.rebalance(args.recut_fraction)
                if args.report_every and done % args.report_every == 0:
                    per = group.num_particles()[1]
                    out["balance"].append({"step": done, "max_over_mean": round(float(per.max() / per.mean()), 4),
                                           "plane_starts": list(group.plane_starts)})
            group.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            out["windows"].append({"ms_per_group_step": round(ms, 4), "ms_per_slab_step": round(ms / world, 4)})
        per = group.num_particles()[1]
        out["residents"] = [int(v) for v in per]
        out["resident_max_over_mean"] = round(float(per.max() / per.mean()), 4)
        out["plane_starts_end"] = list(group.plane_starts)
        out["migrated"], out["max_hop"] = group.migration_stats()
        out["rebins_per_slab"] = [h.timings()["rebin"]["performed"] for h in group.handles]
        group.close()
        print(json.dumps(out))
        return
    group = LocalSlabGroup(grid, starts, [int(c * 1.3) + 4096 for c in share])
    for h in group.handles:
        h.generate_particles(kind, n, seed=seed)
        h.set_domain_particles(n)
    group.set_uniforms(**uni)
    for _ in range(args.warmup):
        group.step()
    group.finish()
    for h in group.handles:
        h.reset_timings()
    if args.profile_slab >= 0:
        group.handles[args.profile_slab].set_profiling(1)
    out["windows"] = []
    for _ in range(args.windows):
        moved0 = group.migrated
        t0 = time.perf_counter()
        for _ in range(args.steps):
            group.step()
        group.finish()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        out["windows"].append({"ms_per_group_step": round(ms, 4), "ms_per_slab_step": round(ms / world, 4),
                               "particle_substeps_per_s_per_gpu_equivalent": round(n / world / (ms / world) * 1e3, 1),
                               "migrated_per_step": round((group.migrated - moved0) / args.steps, 1)})
    timings = [h.timings() for h in group.handles]
    total_steps = args.steps * args.windows
    out["rebins_per_slab"] = [t["rebin"]["performed"] for t in timings]
    out["stragglers_per_step_per_slab"] = [round(t["rebin"]["stragglers"] / total_steps, 1) for t in timings]
    out["residents"] = group.residents()
    if args.profile_slab >= 0:
        t = timings[args.profile_slab]
        out["profiled_slab"] = {"slab": args.profile_slab,
                                "kernel_avg_ms": {k: round(v["ms"] / v["launches"], 4) for k, v in t.items() if v.get("launches")}}
    group.close()
    if args.single:
        h = Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
        h.generate_particles(kind, n, seed=seed)
        h.set_uniforms(**uni)
        h.step(args.warmup, 0.0)
        h.synchronize()
        t0 = time.perf_counter()
        h.step(total_steps, 0.0)
        h.synchronize()
        ms = (time.perf_counter() - t0) / total_steps * 1e3
        out["single_domain"] = {"ms_per_step": round(ms, 4), "ms_per_slab_share": round(ms / world, 4),
                                "rebins": h.timings()["rebin"]["performed"]}
        h.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
