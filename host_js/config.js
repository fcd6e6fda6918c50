'use strict';
// Tunables with the reference's defaults: the data side of its `conf` singleton
// (src/conf.js:6-46).  The tweakpane GUI, the sensors and the fps graph are browser-only.
class Conf {
    constructor(mobile) {
        this.maxParticles = 8192 * 32;      // src/conf.js:8
        this.particles = 8192 * 10;         // src/conf.js:9
        this.run = true;
        this.noise = 1.0;
        this.gridNoiseStrength = 0.5;
        this.speed = 1;
        this.stiffness = 3.0;
        this.restDensity = 1.0;
        this.density = 1;
        this.dynamicViscosity = 0.1;
        this.gravity = 0;
        this.actualSize = 1;
        this.size = 0.5;
        this.mode = 0;
        this.cursorInteraction = false;
        this.hidePercentage = 0;
        if (mobile) { this.maxParticles = 8192 * 16; this.particles = 4096; }   // src/conf.js:33-36
        this.updateParams();
    }

    updateParams() {                        // src/conf.js:41-46
        const level = Math.max(this.particles / 8192, 1);
        this.actualSize = (1.6 / Math.pow(level, 1 / 3)) * this.size;
        this.restDensity = 0.25 * level * this.density;
    }
}
module.exports = { Conf, conf: new Conf(false) };
