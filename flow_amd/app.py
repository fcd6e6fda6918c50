"""Headless counterpart of the reference's caller: the `App` class (src/app.js:24-166) and the
page's animation loop (app/page.js:104-113), without a window, WebGPU or three.js.

What is kept is what drives the simulator and what consumes its buffers:

* `App.init` creates the simulator and the two renderer views (src/app.js:72-80);
* `App.update(delta, elapsed)` runs the per-frame sequence of src/app.js:142-165: renderer
  views update their instance counts (particleRenderer.js:152-155, pointRenderer.js:32-35),
  the simulator steps, the frame is "rendered" (here: optionally written as a PPM image);
* `App.onMouseMove` turns pointer coordinates into the mouse ray exactly as src/app.js:128-139
  does: a ray from the camera through the pointer, intersected with the plane z = 0.2;
* `run_headless` is the animation loop with a fixed-rate clock in place of THREE.Clock.

The camera is the one OrbitControls leaves after its first update (src/app.js:40-56): placed
at (0, 0, -20), pulled in to maxDistance 3 from the target (0, 0.5, 0.2).  Orbiting by the
user, lights, bloom and the tweakpane GUI are browser-only and have no counterpart.
"""
import math

import numpy as np

from .conf import conf as default_conf
from .simulator import MlsMpmSimulator, Vector3


def _normalize(v):
    return v / np.linalg.norm(v)


class Camera:
    """PerspectiveCamera(20, aspect, 0.01, 5) looking at the OrbitControls target."""

    def __init__(self, aspect=16 / 9):
        self.fov, self.aspect = 20.0, aspect
        self.target = np.array([0.0, 0.5, 0.2])
        offset = np.array([0.0, 0.0, -20.0]) - self.target           # src/app.js:41, 46
        radius = min(np.linalg.norm(offset), 3.0)                      # controls.maxDistance (:52)
        theta = math.atan2(offset[0], offset[2])
        phi = math.acos(max(-1.0, min(1.0, offset[1] / np.linalg.norm(offset))))
        phi = max(0.2 * math.pi, min(0.8 * math.pi, phi))              # :53-54
        if theta < 0:
            theta += 2 * math.pi
        theta = max(0.7 * math.pi, min(1.3 * math.pi, theta))          # :55-56
        s = radius * math.sin(phi)
        self.position = self.target + np.array([s * math.sin(theta), radius * math.cos(phi), s * math.cos(theta)])

    def basis(self):
        forward = _normalize(self.target - self.position)
        right = _normalize(np.cross(forward, np.array([0.0, 1.0, 0.0])))
        up = np.cross(right, forward)
        return forward, right, up

    def ray(self, ndc_x, ndc_y):
        """Raycaster.setFromCamera for a perspective camera: (origin, unit direction)."""
        forward, right, up = self.basis()
        t = math.tan(math.radians(self.fov) / 2)
        direction = _normalize(forward + right * (ndc_x * t * self.aspect) + up * (ndc_y * t))
        return self.position.copy(), direction

    def project(self, points):
        """World points -> (ndc x, ndc y, depth along the view axis)."""
        forward, right, up = self.basis()
        rel = points - self.position
        depth = rel @ forward
        t = math.tan(math.radians(self.fov) / 2)
        safe = np.where(depth > 1e-6, depth, np.inf)
        return (rel @ right) / (safe * t * self.aspect), (rel @ up) / (safe * t), depth


class ParticleRendererView:
    """What ParticleRenderer binds (particleRenderer.js:119-133, 152-155): per instance the
    position scaled by (1, 1, 0.4), a sphere scale from the density, the colour; the mesh sits
    at (-1, 0, 0) with scale 1/64 (:143-145)."""

    def __init__(self, sim, conf):
        self.mlsMpmSim, self.conf = sim, conf
        self.instanceCount = sim.numParticles
        self.size = 1.0

    def update(self):
        self.size = self.conf.actualSize
        self.instanceCount = self.conf.particles

    def instances(self):
        n = min(self.instanceCount, self.mlsMpmSim.numParticles)
        pos = self.mlsMpmSim.getPositions()[:n] * np.array([1.0, 1.0, 0.4], np.float32)
        dens = self.mlsMpmSim.getDensities()[:n, 0]
        scale = self.size * np.clip(dens * 0.4 + 0.5, 0.0, 1.0)
        world = pos.astype(np.float64) / 64.0 + np.array([-1.0, 0.0, 0.0])
        return world, scale / 64.0, self.mlsMpmSim.getColors()[:n]


class PointRendererView:
    """What PointRenderer binds (pointRenderer.js:17-19, 32-35): positions only."""

    def __init__(self, sim, conf):
        self.mlsMpmSim, self.conf = sim, conf
        self.instanceCount = 0

    def update(self):
        self.instanceCount = self.conf.particles

    def instances(self):
        n = min(self.instanceCount, self.mlsMpmSim.numParticles)
        pos = self.mlsMpmSim.getPositions()[:n] * np.array([1.0, 1.0, 0.4], np.float32)
        return pos.astype(np.float64) / 64.0 + np.array([-1.0, 0.0, 0.0])


class App:
    def __init__(self, renderer=None, conf=None, width=1280, height=720, **sim_options):
        self.renderer = renderer                     # signature parity (src/app.js:35-37); unused
        self.conf = conf if conf is not None else default_conf
        self.width, self.height = width, height
        self.sim_options = sim_options
        self.camera = None
        self.mlsMpmSim = self.particleRenderer = self.pointRenderer = None
        self.frames = 0

    def init(self, progressCallback=None):
        """src/app.js:39-126, the parts with a headless meaning."""
        self.camera = Camera(self.width / self.height)
        if progressCallback:
            progressCallback(0.5)
        self.mlsMpmSim = MlsMpmSimulator(self.renderer, conf=self.conf, **self.sim_options)
        self.mlsMpmSim.init()
        self.particleRenderer = ParticleRendererView(self.mlsMpmSim, self.conf)
        self.pointRenderer = PointRendererView(self.mlsMpmSim, self.conf)
        self.mlsMpmSim.particleRenderer = self.particleRenderer          # src/app.js:80
        self.plane_normal, self.plane_constant = np.array([0.0, 0.0, -1.0]), 0.2   # :123
        if progressCallback:
            progressCallback(1.0)
        return self

    def resize(self, width, height):
        self.width, self.height = width, height
        self.camera.aspect = width / height

    def onMouseMove(self, clientX, clientY):
        """src/app.js:128-139."""
        x = (clientX / self.width) * 2 - 1
        y = -(clientY / self.height) * 2 + 1
        origin, direction = self.camera.ray(x, y)
        intersect = np.zeros(3)                      # stays (0, 0, 0) when the ray misses the plane
        denom = float(direction @ self.plane_normal)
        if denom != 0.0:
            t = -(float(origin @ self.plane_normal) + self.plane_constant) / denom
            if t >= 0:
                intersect = origin + direction * t
        self.mlsMpmSim.setMouseRay(Vector3(*origin), Vector3(*direction), Vector3(*intersect))

    def update(self, delta, elapsed, frame_path=None):
        """src/app.js:142-165."""
        self.particleRenderer.update()
        self.pointRenderer.update()
        self.mlsMpmSim.update(delta, elapsed)
        if frame_path:
            self.render_ppm(frame_path)
        self.frames += 1

    def render_ppm(self, path, width=None, height=None):
        """A point-splat picture of what the particle renderer would draw: white background,
        unlit per-particle colour, nearest particle wins.  Returns the image array."""
        width, height = width or self.width, height or self.height
        if self.conf.points:
            world = self.pointRenderer.instances()
            colors = np.zeros((world.shape[0], 3), np.float32)
        else:
            world, _, colors = self.particleRenderer.instances()
        x, y, depth = self.camera.project(world)
        px = np.floor((x * 0.5 + 0.5) * width).astype(np.int64)
        py = np.floor((0.5 - y * 0.5) * height).astype(np.int64)
        ok = (depth > 0.01) & (depth < 5.0) & (px >= 0) & (px < width) & (py >= 0) & (py < height)
        order = np.argsort(-depth[ok])               # far to near: the nearest is written last
        image = np.full((height, width, 3), 255, np.uint8)
        image[py[ok][order], px[ok][order]] = np.clip(colors[ok][order] * 255.0 + 0.5, 0, 255).astype(np.uint8)
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (width, height))
            f.write(image.tobytes())
        return image


def run_headless(app, frames, fps=60.0, script=None, frame_pattern=None):
    """The animation loop of app/page.js:104-113 with a fixed-rate clock: frame k gets
    delta = 1 / fps and elapsed = (k + 1) / fps.  `script(app, k)` runs before frame k and may
    move the pointer (app.onMouseMove), switch modes or edit app.conf, as a user would."""
    for k in range(frames):
        if script:
            script(app, k)
        path = frame_pattern % k if frame_pattern else None
        app.update(1.0 / fps, (k + 1) / fps, frame_path=path)
    app.mlsMpmSim.synchronize()
    return app
