'use strict';
// Headless driver used by the tests: node run_frames.js <out.bin> <maxParticles> <particles> <frames>
// Runs the default scene for <frames> updates (interval 1/30 -> clamped, elapsed = 0.5 * frame)
// and writes the AoS records.
const fs = require('fs');
const { MlsMpmSimulator, Conf } = require('./index');

(async () => {
    const [out, maxParticles, particles, frames] = process.argv.slice(2);
    const conf = new Conf(false);
    conf.maxParticles = Number(maxParticles);
    conf.particles = Number(particles);
    conf.updateParams();
    const sim = new MlsMpmSimulator(null, { conf, rebinInterval: 2 });
    await sim.init();
    fs.writeFileSync(out + '.init', Buffer.from(sim.particleBuffer.floatArray.buffer));
    for (let f = 0; f < Number(frames); f++) await sim.update(1 / 30, 0.5 * f);
    const rec = sim.readAoS();
    fs.writeFileSync(out, Buffer.from(rec.buffer));
    const pos = sim.getPositions();
    fs.writeFileSync(out + '.pos', Buffer.from(pos.buffer));
})().catch((e) => { console.error(e); process.exit(1); });
