'use strict';
// JavaScript host of the MLS-MPM simulator: the class surface of the reference's
// mlsMpmSimulator (src/mls-mpm/mlsMpmSimulator.js, "SIM") -- constructor (SIM:71-73),
// init() (SIM:74-260), update(interval, elapsed) (SIM:2423-2541), setMouseRay (SIM:1539-1546),
// fields numParticles / gridSize / uniforms.<name>.value / fixedPointMultiplier /
// particleBuffer -- over the HIP library, reached through the Node-API addon.  Where the
// reference submits five three.js compute nodes (SIM:2538-2539) this class calls `step`.
const path = require('path');
const { Vector3 } = require('./vector3');
const { StructuredArray, PARTICLE_STRUCT } = require('./structured_array');
const { conf: defaultConf } = require('./config');
const { referenceScene } = require('./particles');
const { ModeHost } = require('./modes');

let addon = null;
function native() {
    if (!addon) addon = require(path.join(__dirname, 'build', 'flowmpm_napi.node'));   // throws if not built
    return addon;
}

const FIELD = { position: 0, velocity: 1, density: 2, color: 3, direction: 4, mass: 5 };
const uniform = (value) => ({ value });

// What the simulator asks of the device side: one handle on one GPU, or (option `slabs`) a slab group --
// the same domain cut into x-slabs, all driven by this thread through one object of the library, on one
// device or one slab per device over RCCL.  Results are bit-identical either way.
class SingleDomain {
    constructor(grid, maxParticles, o) {
        this.handle = native().create({ gridX: grid[0], gridY: grid[1], gridZ: grid[2], maxParticles, device: o.device,
            algorithm: o.algorithm, rebinInterval: o.rebinInterval });
    }
    upload(floatArray, count) { native().uploadParticles(this.handle, floatArray, count); }
    setNumParticles(n) { native().setNumParticles(this.handle, n); }
    setUniforms(u) { native().setUniforms(this.handle, u); }
    setModeUniforms(buf) { native().setModeUniforms(this.handle, buf); }
    step(n, dtime) { native().step(this.handle, n, dtime); }
    field(id, out) { native().downloadField(this.handle, id, out); }
    particles(out) { return native().downloadParticles(this.handle, out); }
    synchronize() { native().synchronize(this.handle); }
}

class SlabDomain {
    // slabs: {planeStarts: [0, ..., gridX], devices?: [...], maxParticlesPerSlab?: n}
    constructor(grid, maxParticles, o) {
        const starts = o.slabs.planeStarts, n = starts.length - 1;
        const cap = o.slabs.maxParticlesPerSlab || maxParticles;
        this.floatArray = null;
        this.live = -1;
        this.stepped = false;
        this.group = native().groupCreate({ gridX: grid[0], gridY: grid[1], gridZ: grid[2], planeStarts: starts,
            maxParticles: new Array(n).fill(cap), devices: o.slabs.devices, rebinInterval: o.rebinInterval });
    }
    // the group holds the LIVE particles only: the records beyond the live count stay in the host array
    upload(floatArray) { this.floatArray = floatArray; }
    setNumParticles(n) {
        if (this.stepped) throw new Error('the particle count of a decomposed domain is fixed once it has stepped');
        native().groupUpload(this.group, this.floatArray, n);
        this.live = n;
    }
    setUniforms(u) { native().groupSetUniforms(this.group, u); }
    setModeUniforms(buf) { native().groupSetModeUniforms(this.group, buf); }
    step(n, dtime) { this.stepped = true; native().groupStep(this.group, n, dtime); }
    field(id, out) { native().groupDownloadField(this.group, id, out); }
    particles(out) {
        out.set(this.floatArray);
        native().groupDownloadParticles(this.group, out.subarray(0, this.live * 28));
        return this.floatArray.length / 28;
    }
    synchronize() { native().groupSynchronize(this.group); }
    info() { return native().groupInfo(this.group); }
}

class MlsMpmSimulator {
    constructor(renderer, options) {
        const o = options || {};
        this.renderer = renderer || null;        // accepted for signature parity, never used
        this.conf = o.conf || defaultConf;
        this.seed = o.seed === undefined ? 0xF10E5EED : o.seed;
        this.device = o.device || 0;
        this.rebinInterval = o.rebinInterval || 64;
        this.algorithm = o.algorithm === undefined ? 1 : o.algorithm;
        this.numParticles = 0;
        this.gridSize = new Vector3(0, 0, 0);
        this.uniforms = {};
        this.fixedPointMultiplier = 1e7;         // SIM:43
        this.mousePos = new Vector3();
        this.mousePosArray = [];
        this.particleBuffer = null;
        this.domain = null;                      // SingleDomain or SlabDomain
        this.slabs = o.slabs || null;
        this.clock = o.clock || null;            // stands in for Date.now() (SIM:1827, 2428)
        this.modes = null;
    }

    get gridMode() { return this.modes ? this.modes.gridMode : false; }

    async init() {
        const maxParticles = this.conf.maxParticles;
        this.gridSize.set(128, 64, 64);                                   // SIM:76
        this.particleBuffer = new StructuredArray(PARTICLE_STRUCT, maxParticles, 'particleData');
        referenceScene(this.particleBuffer, maxParticles, this.gridSize.toArray(), this.seed);   // SIM:114-129
        const u = this.uniforms;
        u.gravityType = uniform(0); u.gravity = uniform(new Vector3());
        u.stiffness = uniform(0); u.restDensity = uniform(0); u.dynamicViscosity = uniform(0); u.noise = uniform(0);
        u.gridSize = uniform(this.gridSize); u.dt = uniform(0.1); u.numParticles = uniform(0);
        u.gridMode = uniform(0); u.frontGravityMode = uniform(0);
        u.mouseRayDirection = uniform(new Vector3()); u.mouseRayOrigin = uniform(new Vector3());
        u.mouseForce = uniform(new Vector3()); u.cursorInteraction = uniform(0); u.hidePercentage = uniform(0.0);
        u.time = uniform(0.0);                   // three.js TSL `time` node (SIM:22, 1318)
        this.modes = new ModeHost(u, this.clock);                         // SIM:154-254
        const o = { device: this.device, algorithm: this.algorithm, rebinInterval: this.rebinInterval, slabs: this.slabs };
        this.domain = new (this.slabs ? SlabDomain : SingleDomain)(this.gridSize.toArray(), maxParticles, o);
        this.domain.upload(this.particleBuffer.floatArray, maxParticles);
    }

    get handle() { return this.domain ? this.domain.handle : null; }

    setMouseRay(origin, direction, pos) {        // SIM:1539-1546, arguments mutated as there
        origin.multiplyScalar(64);
        pos.multiplyScalar(64);
        origin.add(new Vector3(64, 0, 0));
        this.uniforms.mouseRayDirection.value.copy(direction.normalize());
        this.uniforms.mouseRayOrigin.value.copy(origin);
        this.mousePos.copy(pos);
    }

    // SIM:1776-2421.  Image modes (1, 2, 7) take the decoded image {width, height, data[, channels]}.
    // The simulated state and the colours g2p reads stay as they are (see modes.js);
    // getColors() returns the rewritten colours from now on.
    async toggleGridMode(mode, image) {
        const b = this.particleBuffer;
        this.modes.toggle(mode, b.floatArray, b.structSize, b.layout.color.offset, this.conf.particles, image);
    }

    async update(interval, elapsed) {            // SIM:2423-2541
        const c = this.conf, u = this.uniforms;
        this.modes.updateBoostClocks();                                   // SIM:2426-2494
        u.noise.value = c.noise;
        u.gridNoiseStrength.value = c.gridNoiseStrength;
        u.stiffness.value = c.stiffness;
        u.gravityType.value = c.gravity;
        u.gravity.value.set(0, 0, 0);
        u.cursorInteraction.value = c.cursorInteraction ? 1 : 0;
        u.hidePercentage.value = c.hidePercentage;
        u.dynamicViscosity.value = c.dynamicViscosity;
        u.restDensity.value = c.restDensity;
        u.time.value = elapsed;
        if (c.particles !== this.numParticles) {                          // SIM:2513-2522
            this.numParticles = c.particles;
            u.numParticles.value = c.particles;
            this.domain.setNumParticles(c.particles);
        }
        interval = Math.min(interval, 1 / 60);                            // SIM:2524-2528
        const effectiveSpeed = (u.frontGravityMode.value === 1) ? 4.0 : c.speed;
        u.dt.value = interval * 6 * effectiveSpeed;
        this.mousePosArray.push(this.mousePos.clone());                   // SIM:2530-2534
        if (this.mousePosArray.length > 3) this.mousePosArray.shift();
        if (this.mousePosArray.length > 1) {
            u.mouseForce.value.copy(this.mousePosArray[this.mousePosArray.length - 1])
                .sub(this.mousePosArray[0]).divideScalar(this.mousePosArray.length);
        }
        if (c.run) {                                                      // SIM:2537-2540
            this.domain.setUniforms({
                dt: u.dt.value, stiffness: u.stiffness.value, restDensity: u.restDensity.value,
                dynamicViscosity: u.dynamicViscosity.value, noise: u.noise.value, time: u.time.value,
                mouseRayDirection: u.mouseRayDirection.value.toArray(), mouseRayOrigin: u.mouseRayOrigin.value.toArray(),
                mouseForce: u.mouseForce.value.toArray(), cursorInteraction: u.cursorInteraction.value,
                hidePercentage: u.hidePercentage.value, gridMode: u.gridMode.value,
                frontGravityMode: u.frontGravityMode.value,
            });
            this.domain.setModeUniforms(this.modes.pack());
            this.domain.step(1, 0.0);
        }
    }

    // What the reference's renderers bind (src/mls-mpm/particleRenderer.js:119-133,
    // src/mls-mpm/pointRenderer.js:17-19), as Float32Arrays in particle-id order.
    _field(name, width) {
        const out = new Float32Array(this.numParticles * width);
        this.domain.field(FIELD[name], out);
        return out;
    }
    getPositions() { return this._field('position', 3); }
    getVelocities() { return this._field('velocity', 3); }
    getDensities() { return this._field('density', 1); }
    getDirections() { return this._field('direction', 3); }
    // What material.colorNode reads: the device colours until toggleGridMode first rebinds the
    // material to a rewritten buffer (SIM:2376-2384).
    getColors() {
        if (this.modes.renderColors) return this.modes.renderColors.slice(0, this.numParticles * 3);
        return this._field('color', 3);
    }

    // Refreshes particleBuffer.floatArray with all records in the reference's 28-float layout.
    // The colour floats of the host array stay the host's (toggleGridMode rewrites them).
    readAoS() {
        const b = this.particleBuffer, at = b.layout.color.offset;
        const rec = new Float32Array(b.floatArray.length);
        this.domain.particles(rec);
        for (let i = 0; i < rec.length; i += b.structSize)
            for (let a = 0; a < 3; a++) rec[i + at + a] = b.floatArray[i + at + a];
        b.floatArray.set(rec);
        return b.floatArray;
    }
    synchronize() { this.domain.synchronize(); }
}

module.exports = { MlsMpmSimulator, native, FIELD };
