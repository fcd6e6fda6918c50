"""Benchmark of the MLS-MPM step (particle-substeps/sec) on N MI355X GPUs of one node.

A "step" is one reference update(): clearGrid -> p2g1 -> p2g2 -> updateGrid -> g2p
(src/mls-mpm/mlsMpmSimulator.js:2538-2539) over synthetic particles already resident in HBM.

N = 1 : benchmark configuration "16M particles, 256^3 grid, single MI355X".
N > 1 : weak scaling, one rank per GPU: rank r owns an x-slab of a (64*N) x 512 x 512
        domain holding N * 2^24 particles (N = 8 is "128M particles, 512^3 grid"); halo
        planes and migrating particles travel over RCCL through torch.distributed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec; measured copy ceiling is ~6290 GB/s
# Algorithmic bytes (fp32 SoA, every array touched once per kernel that needs it, grid tiles
# perfectly reused on chip): per particle-step and per cell-step.
BYTES = {
    "clearGrid": (0, 32), "p2g1": (60, 16), "p2g2": (56, 16), "updateGrid": (0, 32), "g2p": (96, 16),
}
# flowmpm_step has no clearGrid launch: its updateGrid also zeroes the fixed-point buffers of the step before
# (the grid is double-buffered), so the `updateGrid` launch carries both stages' 64 bytes per cell.


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--algorithm", choices=("tiled", "tiled_unfused", "scatter"), default="tiled")
    ap.add_argument("--rebin-interval", type=int, default=64,
                    help="upper bound of the steps between re-bins; the device re-bins earlier when particles drift")
    ap.add_argument("--particles-log2", type=int, default=24, help="particles per GPU = 2^k")
    ap.add_argument("--particles", type=int, default=0,
                    help="N=1: an exact particle count instead of 2^k (the reference's default scene has 81,920)")
    ap.add_argument("--grid", type=str, default="", help="override the N=1 grid, e.g. 128,128,128")
    ap.add_argument("--scene", choices=("ellipsoid", "dam_break"), default="ellipsoid")
    ap.add_argument("--mode", default="chaos",
                    choices=("chaos", "modes_off", "front_gravity_cylinder", "sphere", "polygon", "color_spheres",
                             "segment", "image_grid", "everything"),
                    help="display mode whose g2p forces run in the timed steps (N=1 only); modes_off runs the "
                         "kernel that holds the mode branches with every mode switched off")
    ap.add_argument("--timing-stride", type=int, default=8,
                    help="HIP events on the kernels of every k-th timed step (1 = every step)")
    ap.add_argument("--noise", type=float, default=0.0,
                    help="the noise uniform (conf.js default: 1.0); the noise field is computed either way")
    ap.add_argument("--rounds", type=int, default=1,
                    help="N=1: time the K steps this many times back to back; the line reports the first as "
                         "`value` and all of them, with median and minimum, under `rounds`")
    ap.add_argument("--steady-windows", type=int, default=3,
                    help="N=1: after the timed K steps, this many further windows of --steady-steps steps each are "
                         "timed in the same process (the `steady` object: the step with its share of re-bins, "
                         "whatever K was); 0 switches it off")
    ap.add_argument("--steady-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4,
                    help="steps of the same workload the CPU oracle is timed on (about 12 s on 16 threads)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo = rehearsal of the multi-rank path on fewer GPUs than ranks (host-staged transport)")
    ap.add_argument("--slab-yz", type=int, default=512, help="y and z extent of the multi-GPU domain")
    ap.add_argument("--slab-x-per-gpu", type=int, default=64)
    ap.add_argument("--recut-every", type=int, default=-1,
                    help="N>1: re-cut the slabs at the particle quantiles every k steps (SlabSimulation.rebalance); "
                         "-1 = 10 for the dam-break scene, whose column collapses through the domain, else never")
    ap.add_argument("--recut-fraction", type=float, default=0.2,
                    help="at most this fraction of a slab's mean particle count changes hands through a cut per re-cut")
    ap.add_argument("--one-slab", action="store_true",
                    help="N=1 through the multi-GPU code path: one 64 x 512 x 512 slab, no neighbour (what one "
                         "rank of the weak-scaling runs computes, without the exchanges)")
    return ap.parse_args()


def scene_uniforms(n_particles, grid_cells, scene, noise=0.0):
    """Benchmark uniforms: rest density set so the mean density is about the rest density
    (the ellipsoid fills 0.268 of the box), dt = 0.1, no mouse force.  `noise` scales the
    noise force; the kernel evaluates the noise field whatever its value."""
    fill = 0.268 if scene == "ellipsoid" else 0.25
    return dict(dt=0.1, stiffness=3.0, rest_density=n_particles / (fill * grid_cells),
                dynamic_viscosity=0.1, noise=noise, time=0.0)


def mode_setup(name, grid, n):
    """(uniform overrides, flowmpm_mode_uniforms) for --mode: the reference's mode geometry
    (mlsMpmSimulator.js:154-254, 1652-2156), which is laid out for its 128 x 64 x 64 box, scaled
    to the benchmark grid."""
    from flow_amd import _lib, modes
    sx = grid[0] / 128.0
    centre = (grid[0] / 2.0, grid[1] / 2.0, grid[2] / 2.0)
    m = _lib.ModeUniforms()
    m.cylinder_radius, m.cylinder_center_x, m.cylinder_center_y = 44.0 * sx, centre[0], centre[1]
    m.sphere_radius, m.color_sphere_radius, m.negative_space_sphere_radius = 25.0 * sx, 7.0 * sx, 17.0 * sx
    m.sphere_center[:] = centre
    m.color_spheres_boost_duration = modes.COLOR_SPHERES_BOOST_DURATION
    m.color_spheres_boost_time = 0.7
    verts = modes.polygon_vertices()
    m.polygon_vertex_count = len(verts)
    for i, (x, y) in enumerate(verts):
        m.polygon_vertices[i][:] = (x * sx, y * grid[1] / 64.0)
    positions, unique = modes.color_sphere_layout()
    for i in range(13):
        if i < len(positions):
            m.color_sphere_positions[i][:] = [positions[i][a] * grid[a] / (128.0, 64.0, 64.0)[a] for a in range(3)]
        m.palette_colors[i][:] = [float(c) for c in unique[min(i, len(unique) - 1)]]
    for k, (sphere_deg, chunk_deg) in enumerate(modes.SEGMENT_ANGLES):
        a, b = np.deg2rad(sphere_deg), np.deg2rad(chunk_deg)
        m.segment_boost_duration[k] = modes.COLOR_SPHERES_BOOST_DURATION
        m.segment_boost_time[k] = 0.7
        m.segment_main_radius[k], m.segment_sphere_radius[k] = 18.0 * sx, 7.0 * sx
        m.segment_sphere_center[k][:] = (centre[0] + 40 * sx * np.cos(a), centre[1] + 40 * sx * np.sin(a), centre[2])
        m.segment_negative_center[k][:] = (centre[0] + 20 * sx * np.cos(b), centre[1] + 20 * sx * np.sin(b), centre[2])
    gw, gh, start_x, start_y, spacing = modes.image_grid(3, 2, n)
    m.grid_width, m.grid_height = gw, gh
    m.grid_start_x, m.grid_start_y = start_x * sx, start_y * grid[1] / 64.0
    m.grid_spacing_x = m.grid_spacing_y = spacing * sx
    m.grid_z, m.grid_random_offset, m.grid_noise_strength = centre[2], 0.5, 0.5
    uni = {}
    if name == "front_gravity_cylinder":
        uni["front_gravity_mode"], m.cylinder_mode = 1, 1
    elif name == "sphere":
        m.sphere_containment_mode = 1
    elif name == "polygon":
        m.polygon_containment_mode = 1
    elif name == "color_spheres":
        m.color_spheres_mode = 1
    elif name == "segment":
        m.segment_mode[0] = 1
    elif name == "image_grid":
        uni["grid_mode"] = 1
    elif name == "everything":
        m.cylinder_mode = m.sphere_containment_mode = m.two_color_sphere_mode = 1
        m.polygon_containment_mode = m.color_spheres_mode = 1
        for k in range(9):
            m.segment_mode[k] = 1
    return uni, m


def scripted_mouse(k, grid):
    """The dam-break configuration's scripted mouse ray at step k: a ray along +z through the centre of
    the column whose hit point swings sinusoidally; the mouse force is the difference of the hit points
    two frames apart over the 3-deep history (mlsMpmSimulator.js:2530-2534)."""
    swing = 8.0 * np.sin(0.2 * np.array([k - 2, k], dtype=np.float64))
    return dict(cursor_interaction=1, mouse_ray_direction=(0.0, 0.0, 1.0),
                mouse_ray_origin=(0.125 * grid[0], 0.5 * grid[1], 0.0),
                mouse_force=((swing[1] - swing[0]) / 3.0, 0.0, 0.0))


def make_particles(scene, count, grid, seed, first_id=0):
    from flow_amd.particles import dam_break, reference_scene
    chunk = 1 << 21
    out = np.empty((count, 28), dtype=np.float32)
    fn = reference_scene if scene == "ellipsoid" else dam_break
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        out[lo:hi] = fn(hi - lo, grid, seed=seed, first_id=first_id + lo)
    return out


def cpu_baseline(rec, n, grid, uniforms, steps):
    """The CPU oracle (same algorithm as the reference: one iteration per particle, relaxed
    i32 atomic adds on 16-byte cell records, 112-byte AoS particles), OpenMP over the host
    cores, on a bounded sample: the same particles and grid, `steps` steps."""
    from oracle import binding
    binding.build()
    cores = max(1, min(16, os.cpu_count() or 1))
    orc = binding.Oracle(grid, threads=cores)
    u = binding.make_uniforms(grid, n, **uniforms)
    orc.run(rec, u, 1, 0.0)                       # touch pages, warm caches
    t0 = time.perf_counter()
    orc.run(rec, u, steps, 0.0)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "particle-substeps/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the same workload (%d particles, grid %dx%dx%d), oracle in OpenMP mode"
                      % (steps, n, grid[0], grid[1], grid[2])}


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE doubled as gfx950 needs for coalesced reads, plus
    WRITE_SIZE, each from its own --pmc run), or None when that workload was not profiled."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    return table.get(workload, {}).get(kernel)


def roofline_of(timings, n, cells):
    """Roofline entry of the dominant kernel: the launch with the largest average duration.
    The tiled path runs p2g1 + p2g2 as a mass pass and a summed-momentum pass, which moves work
    (and bytes) from the first launch to the second; the reference's algorithmic bytes are defined
    for the pair, so when one of the two passes is the longest launch the pair is judged together
    as "p2g": 116 N + 32 G bytes over the sum of the two average launch durations.

    Only "p2g" and "g2p" carry a fraction in the per-kernel table: clearGrid and updateGrid visit the
    dirty tiles only and the two p2g launches split the work differently from the reference's stages,
    so the all-cells / per-stage numerators say nothing about those launches one by one."""
    full, avg = {}, {}
    for name, (bp, bc) in BYTES.items():
        t = timings[name]
        if t["launches"] == 0:
            continue
        avg[name] = t["ms"] / t["launches"]
        gbs = (bp * n + bc * cells) / (avg[name] * 1e-3) / 1e9
        full[name] = {"avg_ms": round(avg[name], 4), "algorithmic_GBs": round(gbs, 1),
                      "frac": round(gbs / HBM_PEAK_GBS, 4)}
    if "p2g1" in avg and "p2g2" in avg:
        ms = avg["p2g1"] + avg["p2g2"]
        gbs = (116 * n + 32 * cells) / (ms * 1e-3) / 1e9
        full["p2g"] = {"avg_ms": round(ms, 4), "algorithmic_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    name = max(avg, key=avg.get)
    if name in ("p2g1", "p2g2") and "p2g" in full:
        name = "p2g"
    roof = {"bound": "hbm", "kernel": name, "achieved": full[name]["algorithmic_GBs"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": full[name]["frac"], "traffic": None, "avg_launch_ms": full[name]["avg_ms"]}
    out = {k: (v if k in ("p2g", "g2p") else {"avg_ms": v["avg_ms"]}) for k, v in full.items()}
    if "p2g" in full and "g2p" in full:
        ms = full["p2g"]["avg_ms"] + full["g2p"]["avg_ms"]
        gbs = (212 * n + 48 * cells) / (ms * 1e-3) / 1e9
        out["p2g+g2p"] = {"avg_ms": round(ms, 4), "algorithmic_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return roof, out


def with_sort(p2g_avg_ms, rebin_ms_per_step, n, cells):
    """P2G with the particle sort (the re-bins the device performed) amortised over the steps."""
    ms = p2g_avg_ms + rebin_ms_per_step
    gbs = (116 * n + 32 * cells) / (ms * 1e-3) / 1e9
    return {"avg_ms": round(ms, 4), "algorithmic_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}


def run_single(args):
    import flow_amd
    from flow_amd import _lib
    n = args.particles if args.particles > 0 else 1 << args.particles_log2
    grid = tuple(int(x) for x in args.grid.split(",")) if args.grid else (256, 256, 256)
    cells = grid[0] * grid[1] * grid[2]
    seed = 0xF10E5EED + 3
    uniforms = scene_uniforms(n, cells, args.scene, args.noise)
    algo = {"tiled": _lib.ALGO_TILED, "tiled_unfused": _lib.ALGO_TILED_UNFUSED, "scatter": _lib.ALGO_SCATTER}[args.algorithm]
    # FLOWMPM_BENCH_PAD: extra particle slots, for experiments with the stride between the store's columns
    h = flow_amd.Handle(grid, n + int(os.environ.get("FLOWMPM_BENCH_PAD", "0")), device=0, algorithm=algo,
                        rebin_interval=args.rebin_interval)
    # The scene is generated on the device (flowmpm_generate_particles: the hosts' seeded generator, bit for
    # bit), so that the run is not tens of seconds of numpy around half a second of GPU work.
    kind = _lib.SCENE_ELLIPSOID if args.scene == "ellipsoid" else _lib.SCENE_DAM_BREAK
    h.generate_particles(kind, n, seed=seed, fill=0.25)
    max_per_cell = None
    if args.scene == "dam_break":        # how crowded the column starts out
        ix = np.floor(h.download_field(_lib.FIELD_POSITION)).astype(np.int64)
        max_per_cell = int(np.bincount((ix[:, 0] * grid[1] + ix[:, 1]) * grid[2] + ix[:, 2]).max())
        del ix
    want_cpu = not args.no_cpu_baseline and args.mode == "chaos"
    rec = h.download_particles() if want_cpu else None      # the CPU baseline runs on the same particles
    if args.mode != "chaos":
        mode_uni, mode_record = mode_setup(args.mode, grid, n)
        uniforms = dict(uniforms, **mode_uni)
        h.set_mode_uniforms(mode_record)
        BYTES["g2p"] = (112, 16)       # the mode branches also read colour (12 B) and id (4 B)
    h.set_uniforms(**uniforms)

    def advance(first, count):
        """`count` steps.  The dam-break configuration drives a scripted mouse ray: a ray along +z
        through the centre of the column whose hit point swings sinusoidally, which changes the
        mouse-force uniform every step (mlsMpmSimulator.js:2530-2534); the other scenes keep
        their uniforms and enqueue all steps in one call."""
        if args.scene != "dam_break":
            h.step(count, 0.0)
            return
        for k in range(first, first + count):
            h.set_uniforms(**dict(uniforms, **scripted_mouse(k, grid)))
            h.step(1, 0.0)

    advance(0, args.warmup)
    h.synchronize()
    h.set_profiling(args.timing_stride)
    h.reset_timings()
    t0 = time.perf_counter()
    advance(args.warmup, args.steps)
    h.synchronize()
    elapsed = time.perf_counter() - t0
    timings = h.timings()
    # Optional repeat measurements in the same process (the simulation simply goes on): `value` stays
    # the first timed region; the others are listed so that run-to-run spread can be judged.
    extra_rounds = []
    done = args.warmup + args.steps
    for r in range(1, args.rounds):
        t1 = time.perf_counter()
        advance(done, args.steps)
        h.synchronize()
        extra_rounds.append(n * args.steps / (time.perf_counter() - t1))
        done += args.steps
    # The steady state: further windows in the same process, long enough to hold the re-bins the flow
    # needs, so that the line says what a step costs with its share of the sort whatever K was.
    steady = None
    if args.steady_windows > 0 and args.steady_steps > 0:
        before = h.timings()
        window_values = []
        for _ in range(args.steady_windows):
            t1 = time.perf_counter()
            advance(done, args.steady_steps)
            h.synchronize()
            window_values.append(n * args.steady_steps / (time.perf_counter() - t1))
            done += args.steady_steps
        after = h.timings()
        # The per-kernel averages of the steady state come from one more window in which EVERY step is timed:
        # a re-bin rides on every tenth step or so and costs its step's launches up to half as much again, so
        # every 8-th step sampled can land on those steps far more or less often than their share (0.378 against
        # 0.412 for the P2G fraction of one and the same run), and timing every step costs the windows above 0.8 %.
        h.set_profiling(1)
        before_kernels = h.timings()
        t1 = time.perf_counter()
        advance(done, args.steady_steps)
        h.synchronize()
        kernel_window_value = n * args.steady_steps / (time.perf_counter() - t1)
        after_kernels = h.timings()
        done += args.steady_steps
        steady_steps = args.steady_windows * args.steady_steps
        rebins = (after["rebin"].get("performed") or 0) - (before["rebin"].get("performed") or 0)
        rebin_ms = after["rebin"].get("performed_ms", 0.0) - before["rebin"].get("performed_ms", 0.0)
        med = float(np.median(window_values))
        steady = {"value": round(med, 1), "unit": "particle-substeps/s", "ms_per_step": round(n / med * 1e3, 4),
                  "steps": steady_steps, "windows": [round(v, 1) for v in window_values],
                  "min": round(min(window_values), 1), "rebins": rebins,
                  "rebin_ms_per_step": round(rebin_ms / steady_steps, 4),       # the scan launches only (see kernels.p2g_with_sort)
                  "stragglers_per_step": round(((after["rebin"].get("stragglers") or 0)
                                                - (before["rebin"].get("stragglers") or 0)) / steady_steps, 1)}
        steady_kernels = {name: {"ms": after_kernels[name]["ms"] - before_kernels[name]["ms"],
                                 "launches": after_kernels[name]["launches"] - before_kernels[name]["launches"]} for name in BYTES}
        steady["kernel_avg_ms"] = {name: round(t["ms"] / t["launches"], 4)
                                   for name, t in steady_kernels.items() if t["launches"]}
        # (the flow goes on: the window the averages come from is a later one, with its own throughput)
        steady["kernel_window"] = {"steps": args.steady_steps, "first_step": done - args.steady_steps,
                                   "value": round(kernel_window_value, 1),
                                   "rebins": (after_kernels["rebin"].get("performed") or 0) - (before_kernels["rebin"].get("performed") or 0)}
    h.close()
    first_roof, per_kernel = roofline_of(timings, n, cells)
    # The line's roofline is the steady one whenever the steady windows ran: the timed window follows a
    # fresh binning by a few steps and holds no re-bin, so its launches are the best case; the steady
    # windows hold the flow's re-bins and the spread-out fluid.  The timed window's figure stays on the
    # line as roofline.first_window.
    roof = first_roof
    # (clearGrid has no launch of its own in flowmpm_step: updateGrid zeroes the older fixed-point set)
    if steady is not None and all(steady_kernels[k]["launches"] for k in ("p2g1", "p2g2", "g2p")):
        roof, steady_per_kernel = roofline_of(steady_kernels, n, cells)
        roof["window"] = "steady: the %d steps that follow the timed window and the %d steps of the steady throughput windows, every step timed" % (args.steady_steps, steady_steps)
        roof["first_window"] = {k: first_roof[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
        steady["kernels"] = steady_per_kernel
    else:
        roof["window"] = "timed"
    rebin_in_window = bool(timings["rebin"].get("performed"))
    if "p2g" in per_kernel and timings["rebin"].get("performed_ms") is not None:
        # P2G with the particle sort included.  A re-bin rides on the step's own launches (the mass pass
        # counts, the momentum pass and g2p store to the new slots), so its cost is inside the p2g and
        # g2p averages of the window it happened in; what the device clocks separately is only the one
        # workgroup that scans the tile counts ("scan_ms").  From the steady windows when they ran (the
        # timed window may hold no re-bin); otherwise from the timed window.
        if steady is not None and all(steady_kernels[k]["launches"] for k in ("p2g1", "p2g2")):
            p2g_ms = sum(steady_kernels[k]["ms"] / steady_kernels[k]["launches"] for k in ("p2g1", "p2g2"))
            per_kernel["p2g_with_sort"] = dict(with_sort(p2g_ms, steady["rebin_ms_per_step"], n, cells),
                                               window="steady")
        else:
            per_kernel["p2g_with_sort"] = dict(with_sort(per_kernel["p2g"]["avg_ms"],
                                                         timings["rebin"]["performed_ms"] / args.steps, n, cells),
                                               window="timed")
    workload_key = "%s_%d_%dx%dx%d" % (args.scene, n, grid[0], grid[1], grid[2])
    roof["traffic"] = measured_traffic(roof["kernel"], workload_key)
    # a committed constant of an earlier counter run on this workload, not a measurement of this run
    roof["traffic_source"] = "profiles/pmc_traffic.json" if roof["traffic"] is not None else None
    rebin = timings["rebin"]
    step_bytes = 212 * n + 112 * cells
    line = {
        "metric": "particle-substeps/sec", "value": n * args.steps / elapsed, "unit": "particle-substeps/s",
        "value_steady": steady["value"] if steady is not None else None,
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+i32-fixed-point",
        "data": "synthetic",
        "config": {"workload": "16M particles, 256^3 grid, single MI355X" if (n == 1 << 24 and not args.grid)
                   else "%d particles, grid %dx%dx%d, single MI355X" % (n, grid[0], grid[1], grid[2]),
                   "scene": args.scene, "mode": args.mode, "particles": n, "grid": list(grid),
                   "algorithm": args.algorithm,
                   "rebin_interval": args.rebin_interval, "timing_stride": args.timing_stride, "dt": 0.1,
                   "noise": args.noise,
                   "cursor_interaction": 1 if args.scene == "dam_break" else 0,
                   "max_particles_per_cell_at_start": max_per_cell},
        "roofline": roof,
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes,
                          "achieved_GBs": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                          "frac_of_8TBs": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels": per_kernel,
        # the re-binning sequence is enqueued every step and decides on the device whether to run
        # the device's own clock brackets every re-bin it performs (all steps, not only the sampled ones)
        "rebin_in_window": rebin_in_window,
        "steady": steady,
        "rebin": {"performed": rebin.get("performed"), "scan_ms": round(rebin.get("performed_ms", 0.0), 3),
                  "ms_per_step": round(rebin.get("performed_ms", 0.0) / args.steps, 4),
                  "timed_steps": rebin["launches"],
                  "avg_ms_per_timed_step": round(rebin["ms"] / rebin["launches"], 4) if rebin["launches"] else None},
    }
    if extra_rounds:
        values = [line["value"]] + extra_rounds
        line["rounds"] = {"values": [round(v, 1) for v in values], "median": round(float(np.median(values)), 1),
                          "min": round(min(values), 1)}
    if want_cpu:
        line["cpu_baseline"] = cpu_baseline(rec, n, grid, uniforms, args.cpu_steps)
    return line


class _SingleRank:
    """What run_distributed asks of torch.distributed, for a world of one rank (`--one-slab`)."""

    class ReduceOp:
        MAX = "max"

    def is_initialized(self):
        return False

    def barrier(self):
        pass

    def all_reduce(self, tensor, op=None):
        pass

    def all_gather(self, out_list, tensor):
        out_list[0].copy_(tensor)

    def destroy_process_group(self):
        pass


def run_distributed(args):
    import torch
    import torch.distributed as dist
    from flow_amd import _lib
    from flow_amd.slab import SlabSimulation, plane_starts_from_histogram

    single = args.one_slab and "WORLD_SIZE" not in os.environ
    if single:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist = _SingleRank()                # nothing travels: no process group is needed
        args.dist_backend = "none"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_devices = max(torch.cuda.device_count(), 1)
    local_rank = int(os.environ.get("LOCAL_RANK", rank)) % n_devices
    if args.dist_backend == "nccl" and n_devices < world:
        # RCCL cannot put two ranks on one device: with fewer GPUs than ranks the run is a rehearsal of
        # the decomposition over the host-staged transport, and the line says so (config.transport)
        if rank == 0:
            print("bench.py: %d ranks on %d GPU(s): using the gloo transport (rehearsal, not a scaling point)"
                  % (world, n_devices), file=sys.stderr)
        args.dist_backend = "gloo"
    torch.cuda.set_device(local_rank)
    if args.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        comm_dev = torch.device("cuda", local_rank)
    elif single:
        comm_dev = torch.device("cpu")
    else:
        dist.init_process_group("gloo")
        comm_dev = torch.device("cpu")
    n_per = 1 << args.particles_log2
    n_total = n_per * world
    grid = (args.slab_x_per_gpu * world, args.slab_yz, args.slab_yz)
    cells = grid[0] * grid[1] * grid[2]
    seed = 0xF10E5EED + 4
    uniforms = scene_uniforms(n_total, cells, args.scene, args.noise)
    algo = {"tiled": _lib.ALGO_TILED, "tiled_unfused": _lib.ALGO_TILED_UNFUSED, "scatter": _lib.ALGO_SCATTER}[args.algorithm]

    # The slab boundaries are cut so that every slab starts with about the same particle count, from the
    # per-plane histogram of the whole scene (computed on the device straight from the generator: the same
    # on every rank, so no exchange); then every rank generates the scene on its device and keeps the
    # particles of its own planes.
    kind = _lib.SCENE_ELLIPSOID if args.scene == "ellipsoid" else _lib.SCENE_DAM_BREAK
    hist = _lib.scene_plane_histogram(kind, n_total, grid, seed=seed, fill=0.25, device=local_rank)
    starts = plane_starts_from_histogram(hist, world)
    share = int(hist[starts[rank]:starts[rank + 1]].sum())
    recut = args.recut_every if args.recut_every >= 0 else (10 if args.scene == "dam_break" else 0)
    if world == 1:
        recut = 0
    sim = SlabSimulation(grid, starts, rank, world, max_particles=max(int(n_per * 2.5), share + share // 4),
                         device=local_rank, algorithm=algo, rebin_interval=args.rebin_interval, dist=dist,
                         torch_device=torch.device("cuda", local_rank), slab_max_nx=grid[0] if recut else 0)
    sim.generate_initial_particles(kind, n_total, seed=seed, fill=0.25)
    sim.set_uniforms(**uniforms)
    steps_done = 0

    def advance(count):
        nonlocal steps_done
        for _ in range(count):
            sim.step()
            steps_done += 1
            if recut and steps_done % recut == 0:
                sim.rebalance(args.recut_fraction)

    advance(args.warmup)
    sim.synchronize()
    handle = getattr(sim.backend, "handle", None)
    n_start = sim.resident()
    migrated_at_start = sim.migrated_out
    if handle is not None:
        handle.set_profiling(args.timing_stride)
        handle.reset_timings()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    advance(args.steps)
    sim.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    resident = torch.tensor([sim.resident()], dtype=torch.int64, device=comm_dev)
    gathered = [torch.zeros_like(resident) for _ in range(world)]
    dist.all_gather(gathered, resident)
    # load balance: every rank's own kernel time per step (sum of the sampled per-launch averages)
    timings = handle.timings() if handle is not None else None
    busy = 0.0
    if timings is not None:
        busy = sum(t["ms"] / t["launches"] for name, t in timings.items()
                   if name in BYTES and t["launches"])
    rebins = timings["rebin"]["performed"] if timings is not None else 0
    busy_t = torch.tensor([busy, rebins, sim.migrated_out - migrated_at_start], dtype=torch.float64, device=comm_dev)
    busy_all = [torch.zeros_like(busy_t) for _ in range(world)]
    dist.all_gather(busy_all, busy_t)
    line = None
    if rank == 0:
        per_rank = [int(g.item()) for g in gathered]
        per_rank_busy = [round(float(b[0].item()), 4) for b in busy_all]
        # rank 0's own launches (HIP events on its stream, sampled like the single-GPU run),
        # against the algorithmic bytes of the particles and cells that rank holds
        roof, per_kernel = None, None
        if handle is not None:
            local_cells = (starts[1] - starts[0]) * grid[1] * grid[2]
            roof, per_kernel = roofline_of(timings, (n_start + per_rank[0]) // 2, local_cells)
            roof["scope"] = "rank 0: %d planes, about %d particles" % (starts[1] - starts[0], (n_start + per_rank[0]) // 2)
        line = {
            "metric": "particle-substeps/sec", "value": n_total * args.steps / elapsed,
            "unit": "particle-substeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+i32-fixed-point", "data": "synthetic",
            "config": {"workload": "%d particles, %dx%dx%d grid, %d x-slabs over %d MI355X (%d particles per GPU)"
                                   % (n_total, grid[0], grid[1], grid[2], world, world, n_per),
                       "transport": args.dist_backend,
                       "scene": args.scene, "particles": n_total, "grid": list(grid), "algorithm": args.algorithm,
                       "noise": args.noise, "timing_stride": args.timing_stride,
                       "rebin_interval": args.rebin_interval, "parallelism": "slab%d" % world,
                       "plane_starts": [int(s) for s in starts], "plane_starts_end": [int(s) for s in sim.plane_starts],
                       "recut_every": recut, "recuts": sim.recuts, "resident_per_rank_end": per_rank,
                       "kernel_ms_per_step_per_rank": per_rank_busy,
                       "rebins_per_rank": [int(b[1].item()) for b in busy_all],
                       "particles_sent_away_per_step_per_rank": [round(float(b[2].item()) / args.steps, 1) for b in busy_all],
                       "kernel_time_imbalance_max_over_mean":
                           round(max(per_rank_busy) / (sum(per_rank_busy) / world), 4) if sum(per_rank_busy) else None},
            "roofline": roof, "kernels": per_kernel, "cpu_baseline": None,
        }
    dist.destroy_process_group()
    return line


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` as a plain command: the ranks are started as a CHILD process
    (torch.distributed.run, one rank per GPU) before this process has imported torch or touched
    the GPU; rank 0's JSON line and the child's exit status are relayed."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=os.getcwd())
    lines = child.stdout.splitlines()
    json_lines = [l for l in lines if l.startswith("{")]
    for l in lines:
        if not (json_lines and l is json_lines[-1]):
            print(l, file=sys.stderr)
    if json_lines:
        print(json_lines[-1])
    sys.stdout.flush()
    if child.returncode == 0 and not json_lines:
        return 1
    return child.returncode


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(world_env or "1")
    if args.gpus > 1 or world > 1 or args.one_slab:
        if world != args.gpus:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)"
                             % (args.gpus, world))
        line = run_distributed(args)
    else:
        line = run_single(args)
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
