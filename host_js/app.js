'use strict';
// Headless counterpart of the reference's caller: the App class (src/app.js:24-166) and the
// page's animation loop (app/page.js:104-113) without a window, WebGPU or three.js.  Kept:
// what drives the simulator (init, update, onMouseMove -> setMouseRay) and what consumes its
// buffers (the two renderers' per-instance inputs).  The camera is the one OrbitControls
// leaves after its first update (src/app.js:40-56): placed at (0, 0, -20), pulled in to
// maxDistance 3 from the target (0, 0.5, 0.2).  Lights, bloom, orbiting and the GUI are
// browser-only.
const fs = require('fs');
const { Vector3 } = require('./vector3');
const { conf: defaultConf } = require('./config');
const { MlsMpmSimulator } = require('./simulator');

const sub = (a, b) => [a[0] - b[0], a[1] - b[1], a[2] - b[2]];
const add = (a, b) => [a[0] + b[0], a[1] + b[1], a[2] + b[2]];
const scale = (a, s) => [a[0] * s, a[1] * s, a[2] * s];
const dot = (a, b) => a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
const normalize = (a) => scale(a, 1 / Math.sqrt(dot(a, a)));
const clamp = (x, lo, hi) => Math.max(lo, Math.min(hi, x));

class Camera {                                   // PerspectiveCamera(20, aspect, 0.01, 5)
    constructor(aspect) {
        this.fov = 20; this.aspect = aspect;
        this.target = [0, 0.5, 0.2];
        const offset = sub([0, 0, -20], this.target);                 // src/app.js:41, 46
        const len = Math.sqrt(dot(offset, offset));
        const radius = Math.min(len, 3.0);                            // controls.maxDistance (:52)
        let theta = Math.atan2(offset[0], offset[2]);
        const phi = clamp(Math.acos(clamp(offset[1] / len, -1, 1)), 0.2 * Math.PI, 0.8 * Math.PI);
        if (theta < 0) theta += 2 * Math.PI;
        theta = clamp(theta, 0.7 * Math.PI, 1.3 * Math.PI);           // :55-56
        const s = radius * Math.sin(phi);
        this.position = add(this.target, [s * Math.sin(theta), radius * Math.cos(phi), s * Math.cos(theta)]);
    }
    basis() {
        const forward = normalize(sub(this.target, this.position));
        const right = normalize(cross(forward, [0, 1, 0]));
        return { forward, right, up: cross(right, forward) };
    }
    ray(x, y) {                                  // Raycaster.setFromCamera
        const { forward, right, up } = this.basis();
        const t = Math.tan((this.fov * Math.PI / 180) / 2);
        const direction = normalize(add(forward, add(scale(right, x * t * this.aspect), scale(up, y * t))));
        return { origin: this.position.slice(), direction };
    }
}

// What ParticleRenderer binds (particleRenderer.js:119-133, 152-155).
class ParticleRendererView {
    constructor(sim, conf) { this.mlsMpmSim = sim; this.conf = conf; this.instanceCount = sim.numParticles; this.size = 1; }
    update() { this.size = this.conf.actualSize; this.instanceCount = this.conf.particles; }
    instances() {
        const n = Math.min(this.instanceCount, this.mlsMpmSim.numParticles);
        return { count: n, positions: this.mlsMpmSim.getPositions(), densities: this.mlsMpmSim.getDensities(),
            colors: this.mlsMpmSim.getColors(), positionScale: [1, 1, 0.4], size: this.size,
            objectPosition: [-1, 0, 0], objectScale: 1 / 64 };
    }
}

// What PointRenderer binds (pointRenderer.js:17-19, 32-35).
class PointRendererView {
    constructor(sim, conf) { this.mlsMpmSim = sim; this.conf = conf; this.instanceCount = 0; }
    update() { this.instanceCount = this.conf.particles; }
    instances() {
        return { count: Math.min(this.instanceCount, this.mlsMpmSim.numParticles),
            positions: this.mlsMpmSim.getPositions(), positionScale: [1, 1, 0.4] };
    }
}

class App {
    constructor(renderer, options) {
        const o = options || {};
        this.renderer = renderer || null;
        this.conf = o.conf || defaultConf;
        this.width = o.width || 1280; this.height = o.height || 720;
        this.simOptions = o.sim || {};
        this.frames = 0;
    }

    async init(progressCallback) {               // src/app.js:39-126
        this.camera = new Camera(this.width / this.height);
        if (progressCallback) await progressCallback(0.5);
        this.mlsMpmSim = new MlsMpmSimulator(this.renderer, Object.assign({ conf: this.conf }, this.simOptions));
        await this.mlsMpmSim.init();
        this.particleRenderer = new ParticleRendererView(this.mlsMpmSim, this.conf);
        this.pointRenderer = new PointRendererView(this.mlsMpmSim, this.conf);
        this.mlsMpmSim.particleRenderer = this.particleRenderer;      // :80
        this.plane = { normal: [0, 0, -1], constant: 0.2 };           // :123
        if (progressCallback) await progressCallback(1.0, 100);
    }

    resize(width, height) { this.width = width; this.height = height; this.camera.aspect = width / height; }

    onMouseMove(event) {                         // src/app.js:128-139
        const x = (event.clientX / this.width) * 2 - 1, y = -(event.clientY / this.height) * 2 + 1;
        const { origin, direction } = this.camera.ray(x, y);
        let intersect = [0, 0, 0];               // stays (0, 0, 0) when the ray misses the plane
        const denom = dot(direction, this.plane.normal);
        if (denom !== 0) {
            const t = -(dot(origin, this.plane.normal) + this.plane.constant) / denom;
            if (t >= 0) intersect = add(origin, scale(direction, t));
        }
        this.mlsMpmSim.setMouseRay(new Vector3(...origin), new Vector3(...direction), new Vector3(...intersect));
    }

    async update(delta, elapsed) {               // src/app.js:142-165
        this.particleRenderer.update();
        this.pointRenderer.update();
        await this.mlsMpmSim.update(delta, elapsed);
        this.frames += 1;
    }

    // Writes the instance positions as a point-splat PPM picture (white background, unlit colour).
    renderPpm(path) {
        const { count, positions, colors } = this.particleRenderer.instances();
        const { forward, right, up } = this.camera.basis();
        const t = Math.tan((this.camera.fov * Math.PI / 180) / 2), w = this.width, h = this.height;
        const image = Buffer.alloc(w * h * 3, 255), zbuf = new Float32Array(w * h).fill(Infinity);
        for (let i = 0; i < count; i++) {
            const world = [positions[3 * i] / 64 - 1, positions[3 * i + 1] / 64, positions[3 * i + 2] * 0.4 / 64];
            const rel = sub(world, this.camera.position), depth = dot(rel, forward);
            if (depth <= 0.01 || depth >= 5) continue;
            const px = Math.floor((dot(rel, right) / (depth * t * this.camera.aspect) * 0.5 + 0.5) * w);
            const py = Math.floor((0.5 - dot(rel, up) / (depth * t) * 0.5) * h);
            if (px < 0 || px >= w || py < 0 || py >= h || depth >= zbuf[py * w + px]) continue;
            zbuf[py * w + px] = depth;
            for (let a = 0; a < 3; a++) image[(py * w + px) * 3 + a] = clamp(Math.floor(colors[3 * i + a] * 255 + 0.5), 0, 255);
        }
        fs.writeFileSync(path, Buffer.concat([Buffer.from('P6\n' + w + ' ' + h + '\n255\n'), image]));
    }
}

// The animation loop of app/page.js:104-113 with a fixed-rate clock: frame k gets
// delta = 1 / fps and elapsed = (k + 1) / fps; script(app, k) runs before frame k.
async function runHeadless(app, frames, fps, script) {
    for (let k = 0; k < frames; k++) {
        if (script) await script(app, k);
        await app.update(1 / fps, (k + 1) / fps);
    }
    app.mlsMpmSim.synchronize();
    return app;
}

module.exports = { App, Camera, ParticleRendererView, PointRendererView, runHeadless };
