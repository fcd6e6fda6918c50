"""The JavaScript host (host_js/) and its Node-API addon.  CPU: the addon loads, exposes the
binding, and the host-side pieces agree with the Python mirror.  GPU: the JS class drives the
library through the frame loop and matches the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import helpers
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
                       'downloadParticles','downloadField','downloadGridI32','downloadGridF32','deviceCount','version',
                       'setModeUniforms','modeUniformsSize','groupCreate','groupUpload','groupGenerate','groupSetUniforms',
                       'groupSetModeUniforms','groupStep','groupSynchronize','groupDownloadParticles','groupDownloadField',
                       'groupInfo','groupSlabGridI32','rcclSelftest'];
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


def test_js_mode_host_agrees_with_python_mirror():
    """host_js/modes.js against flow_amd/modes.py: same toggles on the same scripted clock give
    the same packed mode uniforms and the same rewritten colours."""
    from flow_amd import modes as host_modes
    from flow_amd.simulator import Uniform, Vector3, _Uniforms
    schedule = [6, 9, 13, 17, 4, 8, 18, 5, 0]
    out = _node("""
        const m = require('./host_js');
        const schedule = JSON.parse(process.argv[1]);
        let now = 1000;
        const u = {gridMode: {value: 0}, frontGravityMode: {value: 0}};
        const mh = new m.modes.ModeHost(u, () => now);
        const n = 50, stride = 28, arr = new Float32Array(64 * stride);
        const rows = [];
        for (const mode of schedule) {
            mh.toggle(mode, arr, stride, 24, n, null);
            now += 300;
            mh.updateBoostClocks();
            const colors = [];
            for (let i = 0; i < 64; i++) for (let a = 0; a < 3; a++) colors.push(arr[i * stride + 24 + a]);
            rows.push({pack: Array.from(new Uint32Array(mh.pack())), colors: Array.from(new Uint32Array(new Float32Array(colors).buffer)),
                       flags: [u.gridMode.value, u.frontGravityMode.value]});
        }
        const img = {width: 30, height: 20, channels: 3, data: new Uint8Array(30 * 20 * 3)};
        for (let i = 0; i < img.data.length; i++) img.data[i] = (i * 37 + 11) % 256;
        for (const mode of [7, 1]) {
            mh.toggle(mode, arr, stride, 24, n, img);
            const colors = [];
            for (let i = 0; i < 64; i++) for (let a = 0; a < 3; a++) colors.push(arr[i * stride + 24 + a]);
            rows.push({pack: Array.from(new Uint32Array(mh.pack())), colors: Array.from(new Uint32Array(new Float32Array(colors).buffer)),
                       flags: [u.gridMode.value, u.frontGravityMode.value]});
        }
        console.log(JSON.stringify(rows));
    """, json.dumps(schedule))
    rows = json.loads(out)
    now = [1000.0]
    u = _Uniforms()
    u.gridMode, u.frontGravityMode = Uniform(0), Uniform(0)
    mh = host_modes.ModeHost(u, Uniform, Vector3, clock=lambda: now[0])
    colors = np.zeros((64, 3), np.float32)
    img = ((np.arange(30 * 20 * 3) * 37 + 11) % 256).astype(np.uint8).reshape(20, 30, 3)
    for row, (mode, image) in zip(rows, [(m, None) for m in schedule] + [(7, img), (1, img)]):
        mh.toggle(mode, colors, 50, image=image)
        if image is None:
            now[0] += 300
            mh.update_boost_clocks()
        want = np.frombuffer(bytes(mh.pack()), dtype=np.float32)
        got = np.array(row["pack"], dtype=np.uint32).view(np.float32)
        # cos / sin of the two hosts may differ in the last place of a double
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-38, err_msg="mode %d" % mode)
        flags = np.array(row["pack"], dtype=np.uint32)[:14]
        assert np.array_equal(flags, np.frombuffer(bytes(mh.pack()), dtype=np.uint32)[:14])
        assert row["flags"] == [u.gridMode.value, u.frontGravityMode.value]
        got_colors = np.array(row["colors"], dtype=np.uint32).view(np.float32).reshape(64, 3)
        np.testing.assert_allclose(got_colors, colors, rtol=1e-6, atol=1e-7, err_msg="colours, mode %d" % mode)


def test_js_ppm_reader_and_image_mode(tmp_path):
    from flow_amd import modes as host_modes
    img = ((np.arange(30 * 20 * 3) * 37 + 11) % 256).astype(np.uint8).reshape(20, 30, 3)
    ppm = tmp_path / "img.ppm"
    ppm.write_bytes(b"P6\n# made by a test\n30 20\n255\n" + img.tobytes())
    out = _node("""
        const m = require('./host_js');
        const p = m.modes.readPpm(process.argv[1]);
        const u = {gridMode: {value: 0}, frontGravityMode: {value: 0}};
        const mh = new m.modes.ModeHost(u, () => 1);
        const arr = new Float32Array(700 * 28);
        mh.toggle(7, arr, 28, 24, 600, process.argv[1]);
        console.log(JSON.stringify({w: p.width, h: p.height, sum: p.data.reduce((a, b) => a + b, 0),
            grid: [u.gridWidth.value, u.gridHeight.value, u.gridStartX.value, u.gridSpacingX.value],
            first: [arr[24], arr[25], arr[26]], last: [arr[599 * 28 + 24], arr[599 * 28 + 25], arr[599 * 28 + 26]]}));
    """, str(ppm))
    d = json.loads(out)
    assert (d["w"], d["h"], d["sum"]) == (30, 20, int(img.sum()))
    gw, gh, sx, _, spacing = host_modes.image_grid(30, 20, 600)
    assert d["grid"] == pytest.approx([gw, gh, sx, spacing])
    want = host_modes.closest_palette(host_modes.image_colors(img, gw, gh, 600))
    assert d["first"] == pytest.approx(list(want[0]), abs=1e-6) and d["last"] == pytest.approx(list(want[599]), abs=1e-6)


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


@pytest.mark.gpu
def test_js_simulator_toggle_grid_mode_matches_oracle(oracle_binding, tmp_path):
    """toggleGridMode + update() of the JS class; the oracle is driven with the uniform values
    the JS host packed for each frame."""
    out = str(tmp_path / "records.bin")
    max_particles, particles = 8192, 6000
    schedule = [[0, 1], [9, 3], [6, 2], [4, 2], [8, 2], [18, 2], [0, 1]]
    r = subprocess.run([NODE, os.path.join(ROOT, "host_js", "run_frames.js"), out, str(max_particles),
                        str(particles), "0", json.dumps(schedule)], capture_output=True, text=True, cwd=ROOT,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ob = oracle_binding
    init = np.fromfile(out + ".init", dtype=np.float32).reshape(max_particles, 28)
    got = np.fromfile(out, dtype=np.float32).reshape(max_particles, 28)
    packs = np.fromfile(out + ".modes", dtype=np.uint8)
    with open(out + ".frames.json") as f:
        frames = json.load(f)
    size = packs.size // len(frames)
    assert size == 940
    grid = (128, 64, 64)
    orc = ob.Oracle(grid, threads=4)
    want = init.copy()
    rest = 0.25 * max(particles / 8192, 1)
    for k, fr in enumerate(frames):
        u = ob.make_uniforms(grid, particles, dt=fr["dt"], stiffness=3.0, rest_density=rest, dynamic_viscosity=0.1,
                             noise=1.0, time=fr["time"], grid_mode=fr["gridMode"],
                             front_gravity_mode=fr["frontGravityMode"])
        m = ob.ModeUniforms.from_buffer_copy(packs[k * size:(k + 1) * size].tobytes())
        orc.step_modes(want, u, m)
    assert_bits_equal(got[:, :24], want[:, :24], "JS host records after the mode schedule")
    from flow_amd.particles import palette_rgb
    rgb = palette_rgb().astype(np.float32)
    colors = np.fromfile(out + ".colors", dtype=np.float32).reshape(particles, 3)
    assert np.array_equal(colors, rgb[np.arange(particles) % 13])
    assert [fr["frontGravityMode"] for fr in frames][6:8] == [1, 1] and frames[6]["dt"] == pytest.approx(0.4)


@pytest.mark.gpu
def test_js_simulator_image_modes_match_oracle(oracle_binding, tmp_path):
    """toggleGridMode(1 | 7 | 2, image.ppm) of the JS class on the GPU.  The oracle is driven by grid
    uniforms the test derives from the reference's lines (helpers.reference_image_grid, SIM:1565-1602),
    not by what the JS host packed; getColors() is checked against SIM:1605-1632, 2196-2266."""
    out = str(tmp_path / "records.bin")
    max_particles, particles = 8192, 6000
    wide, tall = helpers.sample_image(30, 20, 11), helpers.sample_image(17, 40, 5)
    wide_ppm, tall_ppm = str(tmp_path / "wide.ppm"), str(tmp_path / "tall.ppm")
    helpers.write_ppm(wide_ppm, wide)
    helpers.write_ppm(tall_ppm, tall)
    schedule = [[1, 3, wide_ppm], [7, 2, tall_ppm], [2, 2, wide_ppm], [0, 1]]
    r = subprocess.run([NODE, os.path.join(ROOT, "host_js", "run_frames.js"), out, str(max_particles),
                        str(particles), "0", json.dumps(schedule)], capture_output=True, text=True, cwd=ROOT,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    import mode_cases
    ob = oracle_binding
    init = np.fromfile(out + ".init", dtype=np.float32).reshape(max_particles, 28)
    got = np.fromfile(out, dtype=np.float32).reshape(max_particles, 28)
    with open(out + ".frames.json") as f:
        frames = json.load(f)
    grid = (128, 64, 64)
    orc = ob.Oracle(grid, threads=4)
    want = init.copy()
    rest = 0.25 * max(particles / 8192, 1)
    k = 0
    for entry, (mode, count, pixels) in enumerate(((1, 3, wide), (7, 2, tall), (2, 2, wide), (0, 1, None))):
        fields = dict(mode_cases._BASE, grid_noise_strength=0.5)            # conf.gridNoiseStrength, src/conf.js:15
        if pixels is not None:
            fields.update(helpers.reference_image_grid(pixels.shape[1], pixels.shape[0], particles))
            colors = np.fromfile(out + ".colors.%d" % entry, dtype=np.float32).reshape(particles, 3)
            helpers.assert_image_mode_colors(colors, pixels, mode, particles)
        m = mode_cases.fill(ob.ModeUniforms, fields)
        for _ in range(count):
            fr = frames[k]
            assert fr["gridMode"] == (1 if pixels is not None else 0)
            u = ob.make_uniforms(grid, particles, dt=fr["dt"], stiffness=3.0, rest_density=rest,
                                 dynamic_viscosity=0.1, noise=1.0, time=fr["time"], grid_mode=fr["gridMode"])
            orc.step_modes(want, u, m)
            k += 1
    assert_bits_equal(got[:particles, :24], want[:particles, :24], "JS host records after the image modes")


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", ["[0,60,128]", "[0,40,66,90,128]"], ids=["two_slabs", "four_slabs"])
def test_js_simulator_over_a_slab_group_matches_oracle(oracle_binding, tmp_path, slabs):
    """The JavaScript class with the option `slabs`: one Node process drives several slab handles on the one
    GPU through flowmpm_slab_group (halo merges and particle migration inside the library), frame by frame
    through update() -- records, getPositions() and the conservation sums equal the single-domain oracle's."""
    out = str(tmp_path / "records.bin")
    max_particles, particles, frames = 16384, 12000, 12
    env = dict(os.environ, FLOW_SLABS=slabs)
    r = subprocess.run([NODE, os.path.join(ROOT, "host_js", "run_frames.js"), out, str(max_particles),
                        str(particles), str(frames)], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    init = np.fromfile(out + ".init", dtype=np.float32).reshape(max_particles, 28)
    got = np.fromfile(out, dtype=np.float32).reshape(max_particles, 28)
    grid = (128, 64, 64)
    orc = oracle_binding.Oracle(grid, threads=4)
    want = init.copy()
    rest = 0.25 * max(particles / 8192, 1)
    for f in range(frames):
        u = oracle_binding.make_uniforms(grid, particles, dt=(1 / 60) * 6, stiffness=3.0, rest_density=rest,
                                         dynamic_viscosity=0.1, noise=1.0, time=0.5 * f)
        orc.step(want, u)
    assert_bits_equal(got, want, "JS host records after %d frames over %s" % (frames, slabs))
    pos = np.fromfile(out + ".pos", dtype=np.float32).reshape(particles, 3)
    assert_bits_equal(pos, want[:particles, 0:3], "getPositions()")
    with open(out + ".group.json") as f:
        info = json.load(f)
    assert info["slabs"] == len(json.loads(slabs)) - 1 and info["rccl"] is False and info["particles"] == particles
    assert info["conservation"] == [int(v) for v in orc.cell.astype(np.int64).sum(axis=0)] + [particles]
    assert sum(info["perSlab"]) == particles


@pytest.mark.gpu
def test_js_rccl_world_of_one():
    _node("require('./host_js').native().rcclSelftest(0, 65536); console.log('ok');")
