"""Sustained shader clock inside the momentum pass and g2p (diagnostic build, -DFLOWMPM_STAMPS=1):

    tools/variants.sh stamps "-DFLOWMPM_STAMPS=1"
    FLOWMPM_LIBRARY=$PWD/build_variants/stamps.so python tools/clock_stamps.py

Every workgroup stamps s_memtime (one tick per shader cycle) and s_memrealtime (100 MHz) around its
particle loop; clock = shader ticks / real ticks x 100 MHz, median over the workgroups of the last
launch, after about 2 s of back-to-back steps of the benchmark workload.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flow_amd  # noqa: E402
from flow_amd import _lib  # noqa: E402


def main():
    n, grid = 1 << 24, (256, 256, 256)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    h = flow_amd.Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64)
    h.upload_particles(bench.make_particles("ellipsoid", n, grid, 0xF10E5EED + 3))
    h.set_uniforms(**bench.scene_uniforms(n, grid[0] * grid[1] * grid[2], "ellipsoid", noise))
    h.step(steps, 0.0)
    h.synchronize()
    fn = h.L.flowmpm_debug_clock_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64]
    fn.restype = ctypes.c_int
    out = {}
    for kernel, name in ((0, "p2g_momentum"), (1, "g2p")):
        slots = 8192
        buf = np.zeros((slots, 2), dtype=np.uint64)
        _lib.check(h._h, fn(h._h, kernel, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), slots))
        ok = buf[:, 1] > 200                    # workgroups that ran for more than 2 us
        ghz = buf[ok, 0].astype(np.float64) / buf[ok, 1].astype(np.float64) * 0.1
        out[name] = {"workgroups": int(ok.sum()), "clock_GHz_median": round(float(np.median(ghz)), 3),
                     "clock_GHz_p10": round(float(np.percentile(ghz, 10)), 3),
                     "clock_GHz_p90": round(float(np.percentile(ghz, 90)), 3),
                     "loop_us_median": round(float(np.median(buf[ok, 1])) / 100.0, 2)}
    ph = h.L.flowmpm_debug_phase_stamps
    ph.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int64]
    ph.restype = ctypes.c_int
    for kernel, name in ((0, "p2g_momentum"), (1, "g2p")):
        slots = 32768
        buf = np.zeros((slots, 4), dtype=np.uint32)
        _lib.check(h._h, ph(h._h, kernel, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), slots))
        ok = buf[:, 3] > 0
        b = buf[ok].astype(np.float64)
        out[name]["work_items"] = int(ok.sum())
        out[name]["particles_per_item_mean"] = round(float(b[:, 3].mean()), 1)
        # workgroup-time in microseconds, summed over the work items of the launch
        out[name]["sum_us"] = {"set_up": round(float(b[:, 0].sum()) / 100, 1), "loop": round(float(b[:, 1].sum()) / 100, 1),
                               "flush": round(float(b[:, 2].sum()) / 100, 1)}
        out[name]["median_us"] = {"set_up": float(np.median(b[:, 0])) / 100, "loop": float(np.median(b[:, 1])) / 100,
                                  "flush": float(np.median(b[:, 2])) / 100}
        # least-squares loop time = a + b * particles
        A = np.stack([np.ones(b.shape[0]), b[:, 3]], axis=1)
        coef = np.linalg.lstsq(A, b[:, 1] / 100, rcond=None)[0]
        out[name]["loop_us_fit"] = {"per_item": round(float(coef[0]), 2), "per_particle": round(float(coef[1]), 5)}
    h.close()
    print(json.dumps({"steps_before_the_stamped_launch": steps, "noise": noise, "kernels": out}))


if __name__ == "__main__":
    main()
