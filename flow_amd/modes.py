"""Host side of the simulator's display modes: the uniform set-up of toggleGridMode and its
helpers (src/mls-mpm/mlsMpmSimulator.js, "SIM", lines 1548-2421) and the boost clocks of
update() (SIM:2426-2494).

Everything here is plain host arithmetic in doubles, as the reference's JavaScript does it;
the values land in `uniforms.<name>.value` of the simulator and travel to the device through
flowmpm_set_mode_uniforms.  Where the reference draws from Math.random (SIM:2242-2253, 2298)
this module draws from the seeded counter generator of particles.py.

What the reference's toggleGridMode does to the particle buffer needs a careful reading.  It
rewrites the colour floats of the HOST array it filled in init() (the GPU state is never read
back) and wraps that array in a NEW storage buffer (SIM:2337-2341), which it binds to the
render material's colorNode (SIM:2376-2384).  The compute kernels were built against the old
buffer node (structuredArray.js:75-77 resolves `this.buffer` when the node graph is built), so
the simulation keeps its particle state and g2p keeps reading the colours assigned in init();
only the renderer sees the rewritten colours.  `ModeHost.render_colors` is that second buffer's
colour column.
"""
import math
import time

import numpy as np

from . import _lib
from .particles import PALETTE, palette_rgb, uniform01

COLOR_SPHERES_BOOST_DURATION = 1.4        # SIM:34

# Mode numbers of toggleGridMode (SIM:1777-1793).
MODE_CHAOS, MODE_CANVASSER, MODE_PROTEST, MODE_FRONT_GRAVITY, MODE_FRONT_GRAVITY_CYLINDER = 0, 1, 2, 3, 4
MODE_SPHERE, MODE_COLOR_SPHERES, MODE_USA, MODE_POLYGON = 5, 6, 7, 8
MODE_SEGMENT_FIRST = 9                    # 9..17: the nine population segments
MODE_TWO_COLOR_SPHERE = 18

# (sphere angle, negative-space angle) in degrees per population segment
# (SIM:1836, 1878-1879, 1914-1915, 1950-1951, 1986-1987, 2022-2023, 2058-2059, 2094-2095, 2130-2131).
SEGMENT_ANGLES = ((30.0, 30.0), (135.0, 110.0), (270.0, 250.0), (45.0, 65.0), (180.0, 200.0),
                  (300.0, 320.0), (0.0, 20.0), (90.0, 70.0), (210.0, 190.0))

# SIM:1660-1668: vertices of the V outline in its 559 x 302 SVG view box.
_SVG_VERTICES = ((311.096, 125.894), (92.6019, 0.0), (0.834339, 296.632), (261.155, 235.488),
                 (392.251, 199.766), (520.226, 283.002), (558.723, 4.16184))


def polygon_vertices():
    """SIM:1652-1688: the V outline mapped into a centred 120 x 60 window of the 128 x 64 box."""
    sim_w, sim_h = 120, 60
    x0, y0 = (128 - sim_w) / 2, (64 - sim_h) / 2
    return [(x0 + (x / 559) * sim_w, y0 + (y / 302) * sim_h) for x, y in _SVG_VERTICES]


def color_sphere_layout():
    """SIM:1704-1762: (positions, unique palette colours); rows of 3, 4, 3 spheres, 21 apart,
    centred on (64, 32, 32)."""
    rgb = palette_rgb()
    unique, seen = [], set()
    for i, h in enumerate(PALETTE):
        if h not in seen:
            seen.add(h)
            unique.append(rgb[i])
    rows, spacing = (3, 4, 3), 21
    start_y = 32 - (len(rows) - 1) * spacing / 2
    positions = []
    for row, n in enumerate(rows):
        start_x = 64 - (n - 1) * spacing / 2
        for col in range(n):
            if len(positions) < len(unique):
                positions.append((start_x + col * spacing, start_y + row * spacing, 32.0))
    return positions, unique


def image_grid(width, height, num_particles):
    """SIM:1565-1602: grid dimensions and placement for an image of `width` x `height` pixels.
    Returns (gridWidth, gridHeight, startX, startY, spacing)."""
    aspect = width / height
    gh = math.floor(math.sqrt(num_particles / aspect))
    gw = math.floor(gh * aspect)
    while gw * gh < num_particles:
        if gh != 0 and gw / gh < aspect:       # 0 / 0 is NaN in the reference: not less
            gw += 1
        else:
            gh += 1
    spacing = min(120 / gw, 60 / gh)
    return gw, gh, (128 - gw * spacing) / 2, (64 - gh * spacing) / 2, spacing


def image_colors(pixels, gw, gh, num_particles):
    """SIM:1605-1632: one colour per particle, sampled at grid resolution with both axes
    flipped.  `pixels` is [height, width, >= 3] uint8."""
    h, w = pixels.shape[0], pixels.shape[1]
    i = np.arange(num_particles)
    gx, gy = i % gw, i // gw
    keep = gy < gh
    gx, gy = gx[keep], gy[keep]
    nx = gx / (gw - 1) if gw > 1 else np.zeros(gx.shape)
    ny = gy / (gh - 1) if gh > 1 else np.zeros(gy.shape)
    ix = np.floor((1.0 - nx) * (w - 1)).astype(np.int64)
    iy = np.floor((1.0 - ny) * (h - 1)).astype(np.int64)
    return pixels[iy, ix, :3].astype(np.float64) / 255


def closest_palette(colors):
    """SIM:2196-2214: first palette entry at minimal Euclidean RGB distance."""
    rgb = palette_rgb()
    d = np.sqrt(((colors[:, None, :] - rgb[None, :, :]) ** 2).sum(axis=2))
    return rgb[np.argmin(d, axis=1)]


def read_ppm(path):
    """Binary PPM (P6, maxval 255) -> [height, width, 3] uint8.  Stands in for the browser's
    Image + canvas decode (SIM:1550-1559); PNG / JPEG decoding is out of reach here."""
    with open(path, "rb") as f:
        data = f.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    if tokens[0] != b"P6" or int(tokens[3]) != 255:
        raise ValueError("%s: only binary PPM (P6) with maxval 255 is read" % path)
    w, h = int(tokens[1]), int(tokens[2])
    pixels = np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos + 1)
    return pixels.reshape(h, w, 3)


class ModeHost:
    """The mode state of one simulator: uniforms, boost clocks, render colours."""

    def __init__(self, uniforms, uniform_factory, vector3_factory, clock=None, seed=0x0C0104ED):
        self.u = uniforms
        self.clock = clock or (lambda: time.time() * 1000.0)      # Date.now()
        self.seed = seed
        self.toggles = 0
        self.gridMode = False
        self.currentImageMode = None
        self.gridTargetColors = None
        self.polygonVertices = []
        self.colorSpherePositions = []
        self.colorSpheresModeStartTime = 0
        self.segmentModeStartTime = [0] * _lib.SEGMENTS
        self.render_colors = None
        U, V = uniform_factory, vector3_factory
        u = uniforms
        u.cylinderMode = U(0)                                       # SIM:154-161
        u.cylinderRadius, u.cylinderCenterX, u.cylinderCenterY = U(44.0), U(64.0), U(32.0)
        u.sphereContainmentMode, u.polygonContainmentMode = U(0), U(0)
        u.colorSpheresMode, u.twoColorSphereMode = U(0), U(0)
        u.polygonVertexCount = U(0)                                 # SIM:172-176
        u.polygonVertices = [U([0.0, 0.0]) for _ in range(16)]
        u.colorSpheresBoostTime = U(0.0)                            # SIM:177-178
        u.colorSpheresBoostDuration = U(COLOR_SPHERES_BOOST_DURATION)
        for name in _lib.SEGMENT_NAMES:                             # SIM:162-170, 179-234
            setattr(u, "populationSegment%sMode" % name, U(0))
            setattr(u, "populationSegment%sBoostTime" % name, U(0.0))
            setattr(u, "populationSegment%sBoostDuration" % name, U(COLOR_SPHERES_BOOST_DURATION))
            setattr(u, "populationSegment%sMainSphereRadius" % name, U(18.0))
            lower = name[0].lower() + name[1:]
            setattr(u, lower + "SphereRadius", U(7.0))
            setattr(u, lower + "SphereCenter", U(V()))
            setattr(u, self._negative_name(name), U(V()))
        u.negativeSpaceSphereRadius = U(17.0)                       # SIM:215
        u.sphereRadius, u.colorSphereRadius = U(25.0), U(7.0)       # SIM:233, 235
        u.sphereCenterX, u.sphereCenterY, u.sphereCenterZ = U(64.0), U(32.0), U(32.0)
        u.colorSpherePositions = [U(V()) for _ in range(13)]        # SIM:240-245
        u.paletteColors = [U(V()) for _ in range(13)]
        u.gridWidth, u.gridHeight = U(0), U(0)                      # SIM:246-254
        u.gridStartX, u.gridStartY, u.gridSpacingX, u.gridSpacingY = U(0.0), U(0.0), U(0.0), U(0.0)
        u.gridZ, u.gridRandomOffset, u.gridNoiseStrength = U(32.0), U(0.5), U(3.5)

    @staticmethod
    def _negative_name(name):
        # SIM:216-224: the pink one has no suffix
        return "negativeSpaceSphereCenter" + ("" if name == "Pink" else name)

    # ---- set-up helpers ----
    def setupPolygonContainment(self):
        """SIM:1652-1702."""
        self.polygonVertices = polygon_vertices()
        self.u.polygonVertexCount.value = len(self.polygonVertices)
        for i in range(16):
            self.u.polygonVertices[i].value = list(self.polygonVertices[i]) if i < len(self.polygonVertices) \
                else [0.0, 0.0]

    def setupColorSpheres(self):
        """SIM:1704-1774."""
        self.colorSpherePositions, unique = color_sphere_layout()
        for i in range(13):
            if i < len(self.colorSpherePositions):
                self.u.colorSpherePositions[i].value.set(*self.colorSpherePositions[i])
            self.u.paletteColors[i].value.set(*unique[min(i, len(unique) - 1)])

    def loadImageAndSetupGrid(self, pixels, num_particles):
        """SIM:1548-1650 with the decoded pixels handed in ([height, width, >= 3] uint8)."""
        pixels = np.asarray(pixels)
        gw, gh, sx, sy, spacing = image_grid(pixels.shape[1], pixels.shape[0], num_particles)
        self.gridTargetColors = image_colors(pixels, gw, gh, num_particles)
        u = self.u
        u.gridWidth.value, u.gridHeight.value = gw, gh
        u.gridStartX.value, u.gridStartY.value = sx, sy
        u.gridSpacingX.value = u.gridSpacingY.value = spacing

    # ---- toggleGridMode ----
    def toggle(self, mode, host_colors, num_particles, image=None):
        """SIM:1776-2421.  `host_colors` is the colour column of the host particle array
        ([maxParticles, 3] float32 view), rewritten in place as the reference does."""
        u = self.u
        enabled = mode in (MODE_CANVASSER, MODE_PROTEST, MODE_USA)
        self.gridMode = enabled
        u.gridMode.value = 1 if enabled else 0
        u.frontGravityMode.value = 1 if mode in (MODE_FRONT_GRAVITY, MODE_FRONT_GRAVITY_CYLINDER) else 0
        u.cylinderMode.value = 1 if mode == MODE_FRONT_GRAVITY_CYLINDER else 0
        u.sphereContainmentMode.value = 1 if mode == MODE_SPHERE else 0
        u.polygonContainmentMode.value = 1 if mode == MODE_POLYGON else 0
        u.colorSpheresMode.value = 1 if mode == MODE_COLOR_SPHERES else 0
        u.twoColorSphereMode.value = 1 if mode == MODE_TWO_COLOR_SPHERE else 0
        if mode == MODE_POLYGON and not self.polygonVertices:
            self.setupPolygonContainment()
        if mode == MODE_COLOR_SPHERES and not self.colorSpherePositions:
            self.setupColorSpheres()
        if mode == MODE_COLOR_SPHERES:                              # SIM:1826-1831
            self.colorSpheresModeStartTime = self.clock()
            u.colorSpheresBoostTime.value = COLOR_SPHERES_BOOST_DURATION
        else:
            u.colorSpheresBoostTime.value = 0.0
        for k, name in enumerate(_lib.SEGMENT_NAMES):               # SIM:1833-2156
            on = mode == MODE_SEGMENT_FIRST + k
            getattr(u, "populationSegment%sMode" % name).value = 1 if on else 0
            boost = getattr(u, "populationSegment%sBoostTime" % name)
            if not on:
                boost.value = 0.0
                continue
            cx, cy, cz = u.sphereCenterX.value, u.sphereCenterY.value, u.sphereCenterZ.value
            sphere_rad = (SEGMENT_ANGLES[k][0] * math.pi) / 180.0
            chunk_rad = (SEGMENT_ANGLES[k][1] * math.pi) / 180.0
            dist = 40.0
            lower = name[0].lower() + name[1:]
            getattr(u, lower + "SphereCenter").value.set(cx + dist * math.cos(sphere_rad),
                                                          cy + dist * math.sin(sphere_rad), cz)
            near = dist * 0.5
            getattr(u, self._negative_name(name)).value.set(cx + near * math.cos(chunk_rad),
                                                            cy + near * math.sin(chunk_rad), cz)
            self.segmentModeStartTime[k] = self.clock()
            boost.value = COLOR_SPHERES_BOOST_DURATION

        n = int(num_particles)
        draw0 = 8 * self.toggles          # fresh random draws per toggle, as Math.random gives
        self.toggles += 1
        if enabled:                                                 # SIM:2163-2277
            if self.currentImageMode != mode:
                if image is None:
                    raise ValueError("mode %d needs the decoded image (an array or a PPM path)" % mode)
                pixels = read_ppm(image) if isinstance(image, str) else image
                self.loadImageAndSetupGrid(pixels, n)
                self.currentImageMode = mode
            target = self.gridTargetColors
            m = min(n, target.shape[0])
            if mode == MODE_USA:
                colors = closest_palette(target[:m])
            else:
                idx = np.arange(m)
                lerped = uniform01(self.seed, idx, draw0) < 0.6
                u1 = np.maximum(uniform01(self.seed, idx, draw0 + 1), 2.0 ** -33)
                u2 = uniform01(self.seed, idx, draw0 + 2)
                z0 = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
                f = np.clip(np.abs(z0 * 0.25 + 0.1), 0.0, 1.0)[:, None]
                mixed = closest_palette(target[:m]) * f + target[:m] * (1.0 - f)
                colors = np.where(lerped[:, None], mixed, target[:m])
            host_colors[:m] = colors.astype(np.float32)
        else:
            if mode != MODE_FRONT_GRAVITY:                          # SIM:2280-2282
                self.currentImageMode = None
            if mode == MODE_TWO_COLOR_SPHERE:                       # SIM:2285-2310
                c1 = np.array([0x31, 0x72, 0x85]) / 255
                c2 = np.array([0x79, 0xC8, 0xD6]) / 255
                first = uniform01(self.seed, np.arange(n), draw0) < 0.5
                host_colors[:n] = np.where(first[:, None], c1, c2).astype(np.float32)
            else:                                                   # SIM:2312-2334
                rgb = palette_rgb()
                host_colors[:n] = rgb[np.arange(n) % rgb.shape[0]].astype(np.float32)
        self.render_colors = np.array(host_colors, dtype=np.float32)   # SIM:2337-2341

    # ---- per-frame clocks ----
    def update_boost_clocks(self):
        """SIM:2426-2494."""
        u = self.u
        now = None
        if u.colorSpheresMode.value == 1 and self.colorSpheresModeStartTime > 0:
            now = self.clock()
            since = (now - self.colorSpheresModeStartTime) / 1000
            u.colorSpheresBoostTime.value = max(0, COLOR_SPHERES_BOOST_DURATION - since)
        for k, name in enumerate(_lib.SEGMENT_NAMES):
            if getattr(u, "populationSegment%sMode" % name).value == 1 and self.segmentModeStartTime[k] > 0:
                now = self.clock() if now is None else now
                since = (now - self.segmentModeStartTime[k]) / 1000
                getattr(u, "populationSegment%sBoostTime" % name).value = \
                    max(0, COLOR_SPHERES_BOOST_DURATION - since)

    # ---- to the C ABI ----
    def pack(self):
        """The current uniform values as a flowmpm_mode_uniforms record."""
        u = self.u
        m = _lib.ModeUniforms()
        m.cylinder_mode = int(u.cylinderMode.value)
        m.sphere_containment_mode = int(u.sphereContainmentMode.value)
        m.polygon_containment_mode = int(u.polygonContainmentMode.value)
        m.color_spheres_mode = int(u.colorSpheresMode.value)
        m.two_color_sphere_mode = int(u.twoColorSphereMode.value)
        m.cylinder_radius = u.cylinderRadius.value
        m.cylinder_center_x, m.cylinder_center_y = u.cylinderCenterX.value, u.cylinderCenterY.value
        m.sphere_radius = u.sphereRadius.value
        m.sphere_center[:] = (u.sphereCenterX.value, u.sphereCenterY.value, u.sphereCenterZ.value)
        m.polygon_vertex_count = int(u.polygonVertexCount.value)
        for i in range(16):
            m.polygon_vertices[i][:] = u.polygonVertices[i].value
        m.color_spheres_boost_time = u.colorSpheresBoostTime.value
        m.color_spheres_boost_duration = u.colorSpheresBoostDuration.value
        m.color_sphere_radius = u.colorSphereRadius.value
        for i in range(13):
            m.color_sphere_positions[i][:] = u.colorSpherePositions[i].value.toArray()
            m.palette_colors[i][:] = u.paletteColors[i].value.toArray()
        for k, name in enumerate(_lib.SEGMENT_NAMES):
            lower = name[0].lower() + name[1:]
            m.segment_mode[k] = int(getattr(u, "populationSegment%sMode" % name).value)
            m.segment_boost_time[k] = getattr(u, "populationSegment%sBoostTime" % name).value
            m.segment_boost_duration[k] = getattr(u, "populationSegment%sBoostDuration" % name).value
            m.segment_main_radius[k] = getattr(u, "populationSegment%sMainSphereRadius" % name).value
            m.segment_sphere_radius[k] = getattr(u, lower + "SphereRadius").value
            m.segment_sphere_center[k][:] = getattr(u, lower + "SphereCenter").value.toArray()
            m.segment_negative_center[k][:] = getattr(u, self._negative_name(name)).value.toArray()
        m.negative_space_sphere_radius = u.negativeSpaceSphereRadius.value
        m.grid_width, m.grid_height = int(u.gridWidth.value), int(u.gridHeight.value)
        m.grid_start_x, m.grid_start_y = u.gridStartX.value, u.gridStartY.value
        m.grid_spacing_x, m.grid_spacing_y = u.gridSpacingX.value, u.gridSpacingY.value
        m.grid_z = u.gridZ.value
        m.grid_random_offset = u.gridRandomOffset.value
        m.grid_noise_strength = u.gridNoiseStrength.value
        return m
