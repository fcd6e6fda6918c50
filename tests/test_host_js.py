"""The JavaScript host (host_js/) and its Node-API addon.  CPU: the addon loads, exposes the
binding, and the host-side pieces agree with the Python mirror.  GPU: the JS class drives the
library through the frame loop and matches the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import assert_bits_equal
from flow_amd.particles import reference_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "host_js", "build", "flowmpm_napi.node")

pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON),
                                reason="node or the built addon is not available")


def _node(script, *args):
    r = subprocess.run([NODE, "-e", script] + list(args), capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_addon_loads_and_exposes_the_binding():
    out = _node("""
        const m = require('./host_js');
        const names = ['create','uploadParticles','setNumParticles','setUniforms','step','synchronize',
                       'downloadParticles','downloadField','downloadGridI32','downloadGridF32','deviceCount','version'];
        const missing = names.filter((n) => typeof m.native()[n] !== 'function');
        console.log(JSON.stringify({missing, version: m.native().version()}));
    """)
    d = json.loads(out)
    assert d["missing"] == [] and d["version"] >= 100


def test_js_host_pieces_agree_with_python_mirror():
    out = _node("""
        const m = require('./host_js');
        const s = new m.StructuredArray(m.PARTICLE_STRUCT, 300, 'particleData');
        m.referenceScene(s, 300, [128, 64, 64], 0xF10E5EED);
        const c = new m.Conf(false);
        const o = new m.Vector3(0.1, 0.2, 0.3), d = new m.Vector3(0, 0, 2), p = new m.Vector3(0.5, 0.25, -0.2);
        const sim = new m.MlsMpmSimulator(null, {conf: c});
        sim.uniforms.mouseRayDirection = {value: new m.Vector3()};
        sim.uniforms.mouseRayOrigin = {value: new m.Vector3()};
        sim.setMouseRay(o, d, p);
        console.log(JSON.stringify({structSize: s.structSize,
            offsets: Object.keys(s.layout).map((k) => s.layout[k].offset),
            records: Array.from(new Uint32Array(s.floatArray.buffer)),
            restDensity: c.restDensity, particles: c.particles, maxParticles: c.maxParticles,
            origin: o.toArray(), dir: d.toArray(), pos: p.toArray()}));
    """)
    d = json.loads(out)
    assert d["structSize"] == 28 and d["offsets"] == [0, 3, 4, 7, 8, 20, 24]
    want = reference_scene(300, (128, 64, 64), seed=0xF10E5EED)
    assert np.array_equal(np.array(d["records"], dtype=np.uint32), want.reshape(-1).view(np.uint32))
    assert d["restDensity"] == pytest.approx(2.5) and (d["particles"], d["maxParticles"]) == (81920, 262144)
    assert d["origin"] == pytest.approx([70.4, 12.8, 19.2]) and d["dir"] == [0, 0, 1]
    assert d["pos"] == pytest.approx([32, 16, -12.8])


def test_js_host_refuses_to_run_without_a_gpu():
    out = _node("""
        const m = require('./host_js');
        if (m.native().deviceCount() > 0) { console.log('gpu'); process.exit(0); }
        try { m.native().create({gridX: 16, gridY: 16, gridZ: 16, maxParticles: 10}); console.log('created'); }
        catch (e) { console.log('threw ' + e.code); }
    """)
    assert out.strip() in ("gpu", "threw 2")


@pytest.mark.gpu
def test_js_simulator_frame_loop_matches_oracle(oracle_binding, tmp_path):
    out = str(tmp_path / "records.bin")
    max_particles, particles, frames = 16384, 8192, 3
    r = subprocess.run([NODE, os.path.join(ROOT, "host_js", "run_frames.js"), out, str(max_particles),
                        str(particles), str(frames)], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    init = np.fromfile(out + ".init", dtype=np.float32).reshape(max_particles, 28)
    assert_bits_equal(init, reference_scene(max_particles, (128, 64, 64), seed=0xF10E5EED), "JS init records")
    got = np.fromfile(out, dtype=np.float32).reshape(max_particles, 28)
    grid = (128, 64, 64)
    orc = oracle_binding.Oracle(grid, threads=4)
    want = init.copy()
    for f in range(frames):
        u = oracle_binding.make_uniforms(grid, particles, dt=(1 / 60) * 6, stiffness=3.0, rest_density=0.25,
                                         dynamic_viscosity=0.1, noise=1.0, time=0.5 * f)
        orc.step(want, u)
    assert_bits_equal(got, want, "JS host records after %d frames" % frames)
    pos = np.fromfile(out + ".pos", dtype=np.float32).reshape(particles, 3)
    assert_bits_equal(pos, want[:particles, 0:3], "getPositions()")
