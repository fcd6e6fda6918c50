"""GPU parity of g2p's mode branches (SIM:444-1313, 1365-1476) against the oracle, through the
C ABI, bit for bit, for both algorithms; cases in tests/mode_cases.py."""
import hashlib
import json
import os

import numpy as np
import pytest

import flow_amd
import mode_cases
from flow_amd import _lib
import helpers
from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASE = dict(dt=0.1, stiffness=3.0, rest_density=2.5, dynamic_viscosity=0.1, noise=1.0, time=0.375)
ALGOS = [pytest.param(_lib.ALGO_SCATTER, id="scatter"), pytest.param(_lib.ALGO_TILED, id="tiled")]


def _handle(rec, algo, uni, fields, rebin_interval=2):
    h = flow_amd.Handle(mode_cases.GRID, rec.shape[0], algorithm=algo, rebin_interval=rebin_interval)
    h.upload_particles(rec)
    h.set_uniforms(**dict(BASE, **uni))
    h.set_mode_uniforms(mode_cases.fill(_lib.ModeUniforms, fields))
    return h


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", mode_cases.ALL_CASES)
def test_mode_case_matches_oracle(oracle_binding, name, algo):
    ob = oracle_binding
    n, steps = 20000, 4
    uni, fields = mode_cases.case(name)
    rec = mode_cases.particles(n)
    h = _handle(rec, algo, uni, fields)
    h.step(steps, 0.0)
    u = ob.make_uniforms(mode_cases.GRID, n, **dict(BASE, **uni))
    m = mode_cases.fill(ob.ModeUniforms, fields)
    orc = ob.Oracle(mode_cases.GRID, threads=8)
    want = rec.copy()
    for _ in range(steps):
        orc.step_modes(want, u, m)
    assert_bits_equal(h.download_particles(), want, "%s: particles" % name)
    assert np.array_equal(h.download_grid_i32(), orc.cell), "%s: fixed-point grid" % name
    assert_bits_equal(h.download_grid_f32(), orc.cellf, "%s: float grid" % name)
    h.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_golden_mode_digests(algo):
    with open(os.path.join(GOLDEN, "mode_cases.json")) as f:
        gold = json.load(f)
    for name, want in gold["cases"].items():
        uni, fields = mode_cases.case(name)
        rec = mode_cases.particles(gold["particles"])
        h = flow_amd.Handle(mode_cases.GRID, rec.shape[0], algorithm=algo, rebin_interval=1)
        h.upload_particles(rec)
        h.set_uniforms(**dict(gold["uniforms"], **uni))
        h.set_mode_uniforms(mode_cases.fill(_lib.ModeUniforms, fields))
        h.step(gold["steps"], 0.0)
        got = h.download_particles()
        assert hashlib.sha256(got.tobytes()).hexdigest() == want["sha256_particles"], name
        h.close()


def test_modes_through_the_straggler_path(oracle_binding):
    """A long re-binning interval leaves many particles outside their tile's arena; they take
    the straggler kernel, which must apply the same mode branches."""
    ob = oracle_binding
    n, steps = 20000, 6
    uni, fields = mode_cases.case("everything")
    rec = mode_cases.particles(n, seed=9, vel=2.0)
    h = _handle(rec, _lib.ALGO_TILED, uni, fields, rebin_interval=50)
    h.step(steps, 0.0)
    u = ob.make_uniforms(mode_cases.GRID, n, **dict(BASE, **uni))
    m = mode_cases.fill(ob.ModeUniforms, fields)
    orc = ob.Oracle(mode_cases.GRID, threads=8)
    want = rec.copy()
    for _ in range(steps):
        orc.step_modes(want, u, m)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


def test_mode_uniform_changes_between_steps(oracle_binding):
    """The device copy of the mode uniforms follows every setter call: boost clock running
    down, a mode switched, back to the plain path, with the clock advancing."""
    ob = oracle_binding
    n = 8000
    rec = mode_cases.particles(n, seed=4)
    want = rec.copy()
    orc = ob.Oracle(mode_cases.GRID, threads=8)
    h = flow_amd.Handle(mode_cases.GRID, n, algorithm=_lib.ALGO_TILED, rebin_interval=3)
    h.upload_particles(rec)
    schedule = [("segment_3", 1.4), ("segment_3", 0.9), ("segment_3", 0.0), ("color_spheres", 0.7),
                ("none", 0.0), ("front_gravity_cylinder", 0.0), ("image_grid", 0.0), ("none", 0.0)]
    for step, (name, boost) in enumerate(schedule):
        uni, fields = mode_cases.case(name)
        if name.startswith("segment_"):
            fields["segment_boost_time"] = [boost if k == 3 else 0.0 for k in range(9)]
        if name == "color_spheres":
            fields["color_spheres_boost_time"] = boost
        uniforms = dict(BASE, time=0.375 + step / 60, **uni)
        h.set_uniforms(**uniforms)
        h.set_mode_uniforms(mode_cases.fill(_lib.ModeUniforms, fields))
        h.step(1, 0.0)
        orc.step_modes(want, ob.make_uniforms(mode_cases.GRID, n, **uniforms),
                       mode_cases.fill(ob.ModeUniforms, fields))
    assert_bits_equal(h.download_particles(), want, "particles after the schedule")
    h.close()


def test_particle_index_is_the_host_slot_after_rebinning(oracle_binding):
    """Grid mode keys the target on instanceIndex (SIM:446-448); the store is re-sorted by
    tile, so the kernel must use the carried id, not the slot."""
    ob = oracle_binding
    n = 30000
    uni, fields = mode_cases.case("image_grid")
    fields = dict(fields, grid_width=200, grid_height=150, grid_spacing_x=0.6, grid_spacing_y=0.4)
    rec = mode_cases.particles(n, seed=12)
    h = _handle(rec, _lib.ALGO_TILED, uni, fields, rebin_interval=1)
    h.step(5, 1.0 / 60)
    u = ob.make_uniforms(mode_cases.GRID, n, **dict(BASE, **uni))
    m = mode_cases.fill(ob.ModeUniforms, fields)
    orc = ob.Oracle(mode_cases.GRID, threads=8)
    want = rec.copy()
    for k in range(5):
        u.time = np.float32(BASE["time"] + k * (1.0 / 60))
        orc.step_modes(want, u, m)
    assert_bits_equal(h.download_particles(), want, "particles")
    h.close()


def test_mode_errors(oracle_binding):
    h = flow_amd.Handle((16, 16, 16), 16)
    bad = _lib.ModeUniforms()
    bad.polygon_vertex_count = 17
    with pytest.raises(_lib.FlowMpmError) as e:
        h.set_mode_uniforms(bad)
    assert e.value.code == _lib.ERR_INVALID
    h.close()
    slab = flow_amd.Handle((16, 16, 16), 16, slab=(4, 8))
    with pytest.raises(_lib.FlowMpmError) as e:
        slab.set_uniforms(hide_percentage=10.0)        # the slab does not know the domain's particle count
    assert e.value.code == _lib.ERR_UNSUPPORTED
    slab.set_domain_particles(1000)
    slab.set_uniforms(hide_percentage=10.0)
    slab.close()
    with pytest.raises(_lib.FlowMpmError) as e:
        h2 = flow_amd.Handle((16, 16, 16), 16)
        try:
            h2.set_domain_particles(1000)                  # single-GPU handles count their own particles
        finally:
            h2.close()
    assert e.value.code == _lib.ERR_INVALID


def test_simulator_class_toggle_grid_mode(oracle_binding):
    """MlsMpmSimulator.toggleGridMode + update() against the oracle driven with the same
    uniform values (SIM:1776-2421, 2423-2541)."""
    ob = oracle_binding
    from flow_amd.conf import Conf
    from flow_amd.simulator import MlsMpmSimulator
    conf = Conf()
    conf.maxParticles, conf.particles = 8192, 6000
    conf.updateParams()
    now = [50_000.0]
    sim = MlsMpmSimulator(conf=conf, rebin_interval=2, clock=lambda: now[0]).init()
    want = sim.particleBuffer.records().copy()
    colors_at_init = want[:, 24:27].copy()
    orc = ob.Oracle((128, 64, 64), threads=8)
    elapsed = 0.0
    for mode, frames in ((0, 2), (9, 3), (6, 2), (4, 2), (8, 2), (18, 2), (0, 1)):
        sim.toggleGridMode(mode)
        for _ in range(frames):
            now[0] += 16.0
            elapsed += 0.016
            sim.update(0.016, elapsed)
            su = sim.uniforms
            u = ob.make_uniforms((128, 64, 64), conf.particles, dt=su.dt.value, stiffness=su.stiffness.value,
                                 rest_density=su.restDensity.value, dynamic_viscosity=su.dynamicViscosity.value,
                                 noise=su.noise.value, time=su.time.value, grid_mode=su.gridMode.value,
                                 front_gravity_mode=su.frontGravityMode.value)
            m = ob.ModeUniforms.from_buffer_copy(bytes(sim.modes.pack()))
            orc.step_modes(want, u, m)
    got = sim.readAoS()
    assert_bits_equal(got[:conf.particles], want[:conf.particles], "simulated particles")
    assert_bits_equal(got[:, 24:27], colors_at_init, "device colours stay those of init()")
    # the renderer-facing colours are the rewritten ones: after the last toggle (mode 0), the palette cycle
    from flow_amd.particles import palette_rgb
    rgb = palette_rgb().astype(np.float32)
    assert np.array_equal(sim.getColors(), rgb[np.arange(conf.particles) % 13])
    # front gravity runs at speed 4 (SIM:2525-2527)
    sim.toggleGridMode(3)
    sim.update(0.016, elapsed)
    assert sim.uniforms.dt.value == pytest.approx(0.016 * 6 * 4.0)


def test_simulator_class_image_modes(oracle_binding, tmp_path):
    """toggleGridMode(1 | 7 | 2, image) end to end: image -> grid uniforms -> g2p's grid attraction on
    the GPU.  The oracle is driven by uniforms the TEST derives from the reference's lines
    (helpers.reference_image_grid: SIM:1565-1602), not by what the host packed, and the renderer-facing
    colours are checked against the reference's sampling and palette matching (SIM:1605-1632,
    2196-2266)."""
    ob = oracle_binding
    from flow_amd.conf import Conf
    from flow_amd.simulator import MlsMpmSimulator
    conf = Conf()
    conf.maxParticles, conf.particles = 8192, 6000
    conf.updateParams()
    sim = MlsMpmSimulator(conf=conf, rebin_interval=2, clock=lambda: 50_000.0).init()
    want = sim.particleBuffer.records().copy()
    orc = ob.Oracle((128, 64, 64), threads=8)
    wide, tall = helpers.sample_image(30, 20, 11), helpers.sample_image(17, 40, 5)
    ppm = str(tmp_path / "tall.ppm")
    helpers.write_ppm(ppm, tall)
    elapsed = 0.0
    for mode, image, pixels, frames in ((1, wide, wide, 3), (7, ppm, tall, 2), (2, wide, wide, 2), (0, None, None, 1)):
        sim.toggleGridMode(mode, image=image)
        fields = dict(mode_cases._BASE, grid_noise_strength=0.5)        # conf.gridNoiseStrength, src/conf.js:15
        if pixels is not None:
            fields.update(helpers.reference_image_grid(pixels.shape[1], pixels.shape[0], conf.particles))
        m = mode_cases.fill(ob.ModeUniforms, fields)
        for _ in range(frames):
            elapsed += 0.016
            sim.update(0.016, elapsed)
            u = ob.make_uniforms((128, 64, 64), conf.particles, dt=np.float32(0.016 * 6), stiffness=3.0,
                                 rest_density=conf.restDensity, dynamic_viscosity=0.1, noise=1.0,
                                 time=np.float32(elapsed), grid_mode=1 if pixels is not None else 0)
            orc.step_modes(want, u, m)
        if pixels is not None:      # numParticles is set by the first update() (SIM:2513-2522)
            helpers.assert_image_mode_colors(sim.getColors(), pixels, mode, conf.particles)
    got = sim.readAoS()
    assert_bits_equal(got[:conf.particles, :24], want[:conf.particles, :24], "particles after the image modes")
    # the grid attraction has really acted: the same frames without the toggles end elsewhere
    plain = MlsMpmSimulator(conf=conf, rebin_interval=2, clock=lambda: 50_000.0).init()
    for frame in range(8):
        plain.update(0.016, 0.016 * (frame + 1))
    assert not np.array_equal(plain.readAoS()[:conf.particles, 0:3], got[:conf.particles, 0:3])
