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
        this.handle = null;
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
        this.handle = native().create({ gridX: 128, gridY: 64, gridZ: 64, maxParticles, device: this.device,
            algorithm: this.algorithm, rebinInterval: this.rebinInterval });
        native().uploadParticles(this.handle, this.particleBuffer.floatArray, maxParticles);
    }

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
            native().setNumParticles(this.handle, c.particles);
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
            native().setUniforms(this.handle, {
                dt: u.dt.value, stiffness: u.stiffness.value, restDensity: u.restDensity.value,
                dynamicViscosity: u.dynamicViscosity.value, noise: u.noise.value, time: u.time.value,
                mouseRayDirection: u.mouseRayDirection.value.toArray(), mouseRayOrigin: u.mouseRayOrigin.value.toArray(),
                mouseForce: u.mouseForce.value.toArray(), cursorInteraction: u.cursorInteraction.value,
                hidePercentage: u.hidePercentage.value, gridMode: u.gridMode.value,
                frontGravityMode: u.frontGravityMode.value,
            });
            native().setModeUniforms(this.handle, this.modes.pack());
            native().step(this.handle, 1, 0.0);
        }
    }

    // What the reference's renderers bind (src/mls-mpm/particleRenderer.js:119-133,
    // src/mls-mpm/pointRenderer.js:17-19), as Float32Arrays in particle-id order.
    _field(name, width) {
        const out = new Float32Array(this.numParticles * width);
        native().downloadField(this.handle, FIELD[name], out);
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
        native().downloadParticles(this.handle, rec);
        for (let i = 0; i < rec.length; i += b.structSize)
            for (let a = 0; a < 3; a++) rec[i + at + a] = b.floatArray[i + at + a];
        b.floatArray.set(rec);
        return b.floatArray;
    }
    synchronize() { native().synchronize(this.handle); }
}

module.exports = { MlsMpmSimulator, native, FIELD };
