"""CPU tests of the host-side mirror of the reference's class surface and of the C-ABI
library's exported symbols (no compute call: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import flow_amd
from flow_amd import _lib
from flow_amd.conf import Conf
from flow_amd.particles import PALETTE, reference_scene, uniform01
from flow_amd.simulator import MlsMpmSimulator, Vector3
from flow_amd.structured_array import PARTICLE_STRUCT, StructuredArray

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_particle_struct_layout_matches_reference():
    """Offsets the reference's parser yields for its particle struct
    (structuredArray.js:83-119 on mlsMpmSimulator.js:78-86)."""
    sa = StructuredArray(PARTICLE_STRUCT, 4, "particleData")
    assert sa.structSize == 28
    want = {"position": 0, "density": 3, "velocity": 4, "mass": 7, "C": 8, "direction": 20, "color": 24}
    assert {k: v["offset"] for k, v in sa.layout.items()} == want
    assert sa.floatArray.shape == (4 * 28,) and sa.intArray.dtype == np.int32
    assert sa.intArray.ctypes.data == sa.floatArray.ctypes.data


def test_cell_struct_layout():
    sa = StructuredArray({"x": {"type": "int"}, "y": {"type": "int"}, "z": {"type": "int"},
                          "mass": {"type": "int"}}, 2)
    assert sa.structSize == 4
    sa.set(1, "mass", 7)
    assert sa.intArray[7] == 7


def test_structured_array_set_and_errors():
    sa = StructuredArray(PARTICLE_STRUCT, 3)
    sa.set(2, "position", Vector3(1, 2, 3))
    sa.set(2, "mass", 0.5)
    sa.set(1, "color", [0.1, 0.2, 0.3])
    rec = sa.records()
    assert rec[2, 0:3].tolist() == [1, 2, 3] and rec[2, 7] == 0.5
    assert np.allclose(rec[1, 24:27], [0.1, 0.2, 0.3])
    with pytest.raises(KeyError):
        sa.set(0, "nope", 1.0)
    with pytest.raises(ValueError):
        sa.set(0, "position", [1.0])
    with pytest.raises(ValueError):
        StructuredArray({"a": {"type": "quat"}}, 1)


def test_conf_defaults_and_update_params():
    c = Conf()
    assert (c.maxParticles, c.particles) == (262144, 81920)
    assert (c.noise, c.stiffness, c.dynamicViscosity, c.speed) == (1.0, 3.0, 0.1, 1)
    assert c.cursorInteraction is False and c.hidePercentage == 0
    assert c.restDensity == pytest.approx(2.5)                # 0.25 * (81920 / 8192)
    c.particles = 4096
    c.updateParams()
    assert c.restDensity == pytest.approx(0.25)
    m = Conf(mobile=True)
    assert (m.maxParticles, m.particles) == (131072, 4096)


def test_reference_scene_distribution():
    grid = (128, 64, 64)
    rec = reference_scene(5000, grid, seed=1)
    pos = rec[:, 0:3].astype(np.float64)
    u = (pos / np.array(grid) * 2 - 1) / 0.8
    assert (np.sqrt((u * u).sum(1)) <= 1 + 1e-6).all()        # inside the 0.8-scaled unit ball
    assert (rec[:, 7] <= 1).all() and (rec[:, 7] > 0.998 - 1e-6).all()
    assert not rec[:, 4:7].any() and not rec[:, 8:23].any() and not rec[:, 3].any()
    pal = np.array([[int(h[1:3], 16), int(h[3:5], 16), int(h[5:7], 16)] for h in PALETTE]) / 255
    assert np.allclose(rec[:26, 24:27], np.concatenate([pal, pal]), atol=1e-7)
    # counter-based: a sub-range reproduces the same particles
    assert np.array_equal(reference_scene(100, grid, seed=1, first_id=400), rec[400:500])
    assert 0 <= uniform01(3, np.arange(10), 2).min() and uniform01(3, np.arange(10), 2).max() < 1


class _FakeHandle:
    def __init__(self):
        self.uniform_calls, self.steps, self.live = [], 0, None

    def set_uniforms(self, **kw):
        self.uniform_calls.append(kw)

    def set_mode_uniforms(self, modes):
        self.mode_uniforms = modes

    def step(self, n, dtime):
        self.steps += n

    def set_num_particles(self, n):
        self.live = n


def _sim_with_fake_handle():
    sim = MlsMpmSimulator(None, conf=Conf())
    sim.conf.maxParticles = 64          # keep init light; init() only sizes the host array with it
    sim.conf.particles = 32
    sim.conf.updateParams()
    real_handle = flow_amd.simulator.Handle
    try:
        flow_amd.simulator.Handle = lambda *a, **k: type("H", (), {"upload_particles": lambda s, r: None})()
        sim.init()
    finally:
        flow_amd.simulator.Handle = real_handle
    sim.handle = _FakeHandle()
    return sim


def test_update_derives_uniforms_like_the_reference():
    """dt = min(interval, 1/60) * 6 * speed; numParticles follows conf; mouse force is the
    3-deep history difference divided by its length (mlsMpmSimulator.js:2496-2534)."""
    sim = _sim_with_fake_handle()
    assert sim.gridSize.toArray() == [128, 64, 64]
    assert sim.particleBuffer.structSize == 28
    sim.update(0.5, 1.25)
    u = sim.handle.uniform_calls[-1]
    assert u["dt"] == pytest.approx((1 / 60) * 6 * 1)
    assert u["time"] == 1.25 and u["noise"] == 1.0 and u["stiffness"] == 3.0
    assert u["rest_density"] == pytest.approx(0.25) and u["cursor_interaction"] == 0
    assert sim.numParticles == 32 and sim.handle.live == 32 and sim.handle.steps == 1
    sim.update(0.004, 1.3)
    assert sim.handle.uniform_calls[-1]["dt"] == pytest.approx(0.004 * 6)
    assert sim.handle.live == 32

    # mouse: setMouseRay scales and shifts its arguments in place (mlsMpmSimulator.js:1539-1546)
    o, d, p = Vector3(0.1, 0.2, 0.3), Vector3(0, 0, 2), Vector3(0.5, 0.25, -0.2)
    sim.setMouseRay(o, d, p)
    assert o.toArray() == pytest.approx([6.4 + 64, 12.8, 19.2])
    assert d.toArray() == [0, 0, 1] and p.toArray() == pytest.approx([32, 16, -12.8])
    sim.conf.cursorInteraction = True
    sim.update(0.01, 1.4)
    u = sim.handle.uniform_calls[-1]
    assert u["cursor_interaction"] == 1
    # history is now [0, 0, p] -> (p - 0) / 3
    assert u["mouse_force"] == pytest.approx([32 / 3, 16 / 3, -12.8 / 3])
    assert u["mouse_ray_origin"] == pytest.approx([70.4, 12.8, 19.2])
    sim.conf.run = False
    sim.update(0.01, 1.5)
    assert sim.handle.steps == 3


def test_library_exports_every_declared_symbol():
    """include/flowmpm.h and the built library agree, and the binding covers all of it."""
    header = open(os.path.join(ROOT, "include", "flowmpm.h")).read()
    declared = set(re.findall(r"^\s*(?:const char \*|int )\s*(flowmpm_\w+)\s*\(", header, re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), name
    assert L.flowmpm_version() >= 100


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a HIP device the product path refuses to run instead of falling
    back to a CPU path."""
    L = _lib.load()
    n = ctypes.c_int(0)
    rc = L.flowmpm_device_count(ctypes.byref(n))
    if rc == _lib.OK and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.FlowMpmError) as e:
        flow_amd.Handle((16, 16, 16), 100)
    assert e.value.code == _lib.ERR_HIP


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc", ".js")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("test oracle", ""), os.path.join(dirpath, f)


def test_image_mode_set_up_follows_the_reference_lines():
    """flow_amd/modes.py's image -> grid uniforms and colour rewrite against the scalar restatement of
    SIM:1565-1632, 2196-2266 the GPU tests drive the oracle with (tests/helpers.py)."""
    import helpers
    from flow_amd import modes as host_modes
    from flow_amd.simulator import Uniform, Vector3, _Uniforms
    u = _Uniforms()
    u.gridMode, u.frontGravityMode = Uniform(0), Uniform(0)
    mh = host_modes.ModeHost(u, Uniform, Vector3, clock=lambda: 1.0)
    colors = np.zeros((8192, 3), np.float32)
    for mode, img, n in ((1, helpers.sample_image(30, 20, 11), 6000), (7, helpers.sample_image(17, 40, 5), 6000),
                         (2, helpers.sample_image(64, 9, 3), 777), (7, helpers.sample_image(1, 1, 0), 5)):
        mh.currentImageMode = None
        mh.toggle(mode, colors, n, image=img)
        g = helpers.reference_image_grid(img.shape[1], img.shape[0], n)
        assert (u.gridWidth.value, u.gridHeight.value) == (g["grid_width"], g["grid_height"])
        assert (u.gridStartX.value, u.gridStartY.value) == (g["grid_start_x"], g["grid_start_y"])
        assert (u.gridSpacingX.value, u.gridSpacingY.value) == (g["grid_spacing_x"], g["grid_spacing_y"])
        assert u.gridMode.value == 1
        if n > 100:
            helpers.assert_image_mode_colors(mh.render_colors[:n], img, mode, n)
