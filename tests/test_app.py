"""The headless App harness (flow_amd/app.py, host_js/app.js): the caller of the simulator
(src/app.js:24-166, app/page.js:104-113) without a browser."""
import json
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

import flow_amd
from flow_amd.app import App, Camera, run_headless
from flow_amd.conf import Conf
from helpers import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "host_js", "build", "flowmpm_napi.node")


def test_camera_is_where_orbit_controls_leaves_it():
    """src/app.js:40-56: from (0, 0, -20) pulled in to distance 3 of the target."""
    c = Camera(16 / 9)
    assert np.linalg.norm(c.position - c.target) == pytest.approx(3.0)
    offset = np.array([0, 0, -20.0]) - c.target
    assert np.allclose((c.position - c.target) / 3.0, offset / np.linalg.norm(offset))
    origin, direction = c.ray(0.0, 0.0)
    assert np.allclose(direction, (c.target - c.position) / 3.0)
    x, y, depth = c.project(np.array([c.target]))
    assert abs(x[0]) < 1e-12 and abs(y[0]) < 1e-12 and depth[0] == pytest.approx(3.0)


def test_mouse_move_becomes_the_reference_mouse_ray():
    """src/app.js:128-139 + SIM:1539-1546: pointer -> ray -> plane z = 0.2 -> simulator units."""
    app = App(conf=Conf(), width=1000, height=500)
    app.camera = Camera(2.0)
    app.plane_normal, app.plane_constant = np.array([0.0, 0.0, -1.0]), 0.2
    calls = []

    class Sim:
        def setMouseRay(self, origin, direction, pos):
            calls.append((origin.toArray(), direction.toArray(), pos.toArray()))
    app.mlsMpmSim = Sim()
    app.onMouseMove(500, 250)                      # centre of the window
    origin, direction, pos = calls[0]
    assert np.allclose(origin, app.camera.position)
    assert pos[2] == pytest.approx(0.2) and np.allclose(pos[:2], [0.0, 0.5], atol=1e-12)
    app.onMouseMove(900, 100)
    origin, direction, pos = calls[1]
    assert pos[2] == pytest.approx(0.2)
    # the hit point lies on the ray
    t = (np.array(pos) - np.array(origin)) / np.array(direction)
    assert np.allclose(t, t[0])
    assert pos[0] < 0 and pos[1] > 0.5             # right of centre on screen is -x in world space


def test_renderer_views_follow_conf():
    """particleRenderer.js:152-155, pointRenderer.js:32-35."""
    from flow_amd.app import ParticleRendererView, PointRendererView

    class Sim:
        numParticles = 4

        def getPositions(self):
            return np.array([[64, 32, 80], [0, 0, 0], [128, 64, 64], [1, 2, 3]], np.float32)

        def getDensities(self):
            return np.array([[0.0], [1.0], [5.0], [-3.0]], np.float32)

        def getColors(self):
            return np.ones((4, 3), np.float32)
    conf = Conf()
    conf.particles, conf.actualSize = 3, 0.5
    pr, pt = ParticleRendererView(Sim(), conf), PointRendererView(Sim(), conf)
    pr.update(); pt.update()
    world, scale, colors = pr.instances()
    assert world.shape == (3, 3) and np.allclose(world[0], [0.0, 0.5, 0.5])
    assert np.allclose(scale * 64, [0.25, 0.45, 0.5])          # size * clamp(density * 0.4 + 0.5, 0, 1)
    assert np.allclose(pt.instances()[2], [1.0, 1.0, 0.4])


@pytest.mark.gpu
def test_headless_app_matches_oracle(oracle_binding, tmp_path):
    """run_headless with a scripted pointer, cursor interaction on and a mode switch; the oracle
    is driven with the uniform values the frame loop derived."""
    ob = oracle_binding
    conf = Conf()
    conf.maxParticles, conf.particles, conf.cursorInteraction = 8192, 5000, True
    conf.updateParams()
    app = App(conf=conf, width=640, height=360, rebin_interval=4, clock=lambda: 1000.0).init()
    want = app.mlsMpmSim.particleBuffer.records().copy()
    orc = ob.Oracle((128, 64, 64), threads=8)
    log = []

    def script(a, k):
        a.onMouseMove(320 + 60 * math.sin(0.7 * k), 180 + 40 * math.cos(0.5 * k))
        if k == 6:
            a.mlsMpmSim.toggleGridMode(5)

    frames = 10
    for k in range(frames):
        script(app, k)
        app.update(1 / 60, (k + 1) / 60, frame_path=str(tmp_path / "f.ppm") if k == frames - 1 else None)
        su = app.mlsMpmSim.uniforms
        u = ob.make_uniforms((128, 64, 64), conf.particles, dt=su.dt.value, stiffness=su.stiffness.value,
                             rest_density=su.restDensity.value, dynamic_viscosity=su.dynamicViscosity.value,
                             noise=su.noise.value, time=su.time.value, cursor_interaction=1,
                             mouse_ray_direction=su.mouseRayDirection.value.toArray(),
                             mouse_ray_origin=su.mouseRayOrigin.value.toArray(),
                             mouse_force=su.mouseForce.value.toArray())
        m = ob.ModeUniforms.from_buffer_copy(bytes(app.mlsMpmSim.modes.pack()))
        orc.step_modes(want, u, m)
        log.append(su.mouseForce.value.toArray())
    got = app.mlsMpmSim.readAoS()
    assert_bits_equal(got[:conf.particles, :24], want[:conf.particles, :24], "particles after the frame loop")
    assert any(any(abs(c) > 0 for c in f) for f in log), "the scripted pointer must produce a mouse force"
    data = (tmp_path / "f.ppm").read_bytes()
    assert data.startswith(b"P6\n640 360\n255\n")
    image = np.frombuffer(data[len(b"P6\n640 360\n255\n"):], np.uint8).reshape(360, 640, 3)
    drawn = (image != 255).any(axis=2).sum()
    assert 100 < drawn < 640 * 360 / 2, "particles must show up in the frame"
    assert app.particleRenderer.instanceCount == 5000 and app.frames == frames


@pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is not available")
def test_js_camera_and_mouse_ray_agree_with_python():
    r = subprocess.run([NODE, "-e", """
        const m = require('./host_js');
        const app = new m.App(null, {width: 1000, height: 500});
        app.camera = new m.Camera(2.0); app.plane = {normal: [0, 0, -1], constant: 0.2};
        const calls = [];
        app.mlsMpmSim = {setMouseRay: (o, d, p) => calls.push([o.toArray(), d.toArray(), p.toArray()])};
        app.onMouseMove({clientX: 900, clientY: 100});
        console.log(JSON.stringify({position: app.camera.position, call: calls[0]}));
    """], capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    app = App(conf=Conf(), width=1000, height=500)
    app.camera = Camera(2.0)
    app.plane_normal, app.plane_constant = np.array([0.0, 0.0, -1.0]), 0.2
    calls = []
    app.mlsMpmSim = type("S", (), {"setMouseRay": lambda self, o, d_, p: calls.append(
        [o.toArray(), d_.toArray(), p.toArray()])})()
    app.onMouseMove(900, 100)
    assert np.allclose(d["position"], app.camera.position)
    assert np.allclose(np.array(d["call"]), np.array(calls[0]), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is not available")
def test_js_headless_app_matches_oracle(oracle_binding, tmp_path):
    """host_js/app.js: runHeadless with a scripted pointer; the oracle replays the uniform
    values the JS frame loop derived."""
    ob = oracle_binding
    out = str(tmp_path / "js_app")
    r = subprocess.run([NODE, "-e", """
        const fs = require('fs');
        const m = require('./host_js');
        (async () => {
            const out = process.argv[1];
            const conf = new m.Conf(false);
            conf.maxParticles = 8192; conf.particles = 5000; conf.cursorInteraction = true; conf.updateParams();
            const app = new m.App(null, {conf, width: 640, height: 360, sim: {rebinInterval: 4, clock: () => 1000}});
            await app.init();
            fs.writeFileSync(out + '.init', Buffer.from(app.mlsMpmSim.particleBuffer.floatArray.buffer));
            const frames = [];
            await m.runHeadless(app, 8, 60, async (a, k) => {
                if (k > 0) {                     // the uniforms frame k - 1 ran with
                    const u = a.mlsMpmSim.uniforms;
                    frames.push({dt: u.dt.value, time: u.time.value, rest: u.restDensity.value,
                        dir: u.mouseRayDirection.value.toArray(), origin: u.mouseRayOrigin.value.toArray(),
                        force: u.mouseForce.value.toArray()});
                }
                a.onMouseMove({clientX: 320 + 60 * Math.sin(0.7 * k), clientY: 180 + 40 * Math.cos(0.5 * k)});
            });
            const u = app.mlsMpmSim.uniforms;
            frames.push({dt: u.dt.value, time: u.time.value, rest: u.restDensity.value,
                dir: u.mouseRayDirection.value.toArray(), origin: u.mouseRayOrigin.value.toArray(),
                force: u.mouseForce.value.toArray()});
            fs.writeFileSync(out + '.rec', Buffer.from(app.mlsMpmSim.readAoS().buffer));
            app.renderPpm(out + '.ppm');
            console.log(JSON.stringify({frames, count: app.particleRenderer.instanceCount}));
        })().catch((e) => { console.error(e); process.exit(1); });
    """, out], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout)
    assert d["count"] == 5000 and len(d["frames"]) == 8
    want = np.fromfile(out + ".init", dtype=np.float32).reshape(8192, 28)
    orc = ob.Oracle((128, 64, 64), threads=8)
    for fr in d["frames"]:
        u = ob.make_uniforms((128, 64, 64), 5000, dt=fr["dt"], stiffness=3.0, rest_density=fr["rest"],
                             dynamic_viscosity=0.1, noise=1.0, time=fr["time"], cursor_interaction=1,
                             mouse_ray_direction=fr["dir"], mouse_ray_origin=fr["origin"], mouse_force=fr["force"])
        orc.step(want, u)
    got = np.fromfile(out + ".rec", dtype=np.float32).reshape(8192, 28)
    assert_bits_equal(got[:5000, :24], want[:5000, :24], "JS app records")
    assert os.path.getsize(out + ".ppm") == len(b"P6\n640 360\n255\n") + 640 * 360 * 3
