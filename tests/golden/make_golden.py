"""Generates the committed golden vectors from the CPU oracle (oracle/mpm_oracle.c).

The reference holds no fixtures of its own and cannot run here, so these vectors pin
the ORACLE's outputs (parity with the reference itself is unpinned; see
oracle/mpm_oracle.h).  Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding  # noqa: E402
from helpers import moving_particles  # noqa: E402
from flow_amd.particles import reference_scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_case():
    grid, n = (16, 16, 16), 512
    uni = dict(dt=0.1, stiffness=3.0, rest_density=1.0, dynamic_viscosity=0.1, noise=1.0, time=0.375,
               mouse_ray_direction=(0.0, 0.0, 1.0), mouse_ray_origin=(8.0, 8.0, 0.0),
               mouse_force=(0.5, -0.25, 0.125), cursor_interaction=1)
    rec0 = moving_particles(n, grid, seed=11)
    orc = binding.Oracle(grid)
    u = binding.make_uniforms(grid, n, **uni)
    rec = rec0.copy()
    out = {"particles_in": rec0}
    orc.clear_grid()
    orc.p2g1(rec, u)
    out["cell_after_p2g1"] = orc.cell.copy()
    orc.p2g2(rec, u)
    out["cell_after_p2g2"] = orc.cell.copy()
    out["density_after_p2g2"] = rec[:, 3].copy()
    orc.update_grid(u)
    out["cellf_after_update"] = orc.cellf.copy()
    orc.g2p(rec, u)
    out["particles_step1"] = rec.copy()
    orc.run(rec, u, 3, 1.0 / 60)
    out["particles_step4"] = rec.copy()
    out["cell_step4"] = orc.cell.copy()
    np.savez_compressed(os.path.join(HERE, "small_16cube_512.npz"), **out)
    with open(os.path.join(HERE, "small_16cube_512.json"), "w") as f:
        json.dump({"grid": grid, "particles": n, "uniforms": uni, "seed": 11,
                   "steps_after_first": 3, "dtime": 1.0 / 60}, f, indent=1)


def default_scene_digest():
    """Configuration 1 of the benchmark plan: the reference's default scene, one step."""
    grid, n = (128, 64, 64), 81920
    uni = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.0)
    rec = reference_scene(n, grid, seed=0xF10E5EED)
    orc = binding.Oracle(grid)
    orc.step(rec, binding.make_uniforms(grid, n, **uni))
    digest = {
        "grid": grid, "particles": n, "uniforms": uni, "seed": 0xF10E5EED,
        "sha256_cell_i32": hashlib.sha256(orc.cell.tobytes()).hexdigest(),
        "sha256_cellf": hashlib.sha256(orc.cellf.tobytes()).hexdigest(),
        "sha256_particles": hashlib.sha256(rec.tobytes()).hexdigest(),
        "grid_mass_sum_i64": int(orc.cell[:, 3].astype(np.int64).sum()),
    }
    with open(os.path.join(HERE, "default_scene_step1.json"), "w") as f:
        json.dump(digest, f, indent=1)


def noise_vectors():
    pts = np.array([[0.0, 0.0, 0.0], [0.3, 0.45, 0.6], [1.92, 0.96, 0.96], [0.015, 0.5, 0.9],
                    [1.234, 0.777, 0.111]], dtype=np.float32)
    rows = []
    for p in pts:
        for t in (0.0, 1.5, 123.25):
            rows.append({"position": p.tolist(), "speed": t, "time": 0.11,
                         "out_bits": [int(x) for x in binding.tri_noise3d(p, t, 0.11).view(np.uint32)]})
    with open(os.path.join(HERE, "tri_noise.json"), "w") as f:
        json.dump(rows, f, indent=1)


def mode_digests():
    """Three steps per mode case of tests/mode_cases.py: digests of the particle records
    plus the first two records' bits."""
    import mode_cases
    n, steps = 4096, 3
    base = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.375)
    rows = {}
    for name in mode_cases.ALL_CASES:
        uni, fields = mode_cases.case(name)
        u = binding.make_uniforms(mode_cases.GRID, n, **dict(base, **uni))
        m = mode_cases.fill(binding.ModeUniforms, fields)
        rec = mode_cases.particles(n)
        orc = binding.Oracle(mode_cases.GRID)
        for _ in range(steps):
            orc.step_modes(rec, u, m)
        rows[name] = {"sha256_particles": hashlib.sha256(rec.tobytes()).hexdigest(),
                      "first_records_bits": [int(x) for x in rec[:2].view(np.uint32).ravel()]}
    with open(os.path.join(HERE, "mode_cases.json"), "w") as f:
        json.dump({"grid": mode_cases.GRID, "particles": n, "steps": steps, "uniforms": base, "cases": rows},
                  f, indent=1)


if __name__ == "__main__":
    binding.build()
    small_case()
    default_scene_digest()
    noise_vectors()
    mode_digests()
    print("golden vectors written to", HERE)
