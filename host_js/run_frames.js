'use strict';
// Headless driver used by the tests:
//   node run_frames.js <out.bin> <maxParticles> <particles> <frames> [schedule]
// Runs the default scene for <frames> updates (interval 1/30 -> clamped, elapsed = 0.5 * frame)
// and writes the AoS records.  With a schedule (JSON [[mode, frames], ...]) it calls
// toggleGridMode(mode) before each group of frames on a scripted clock (16 ms per frame) and
// also writes, per frame, the packed mode uniforms plus {dt, gridMode, frontGravityMode}
// (<out>.modes, <out>.frames.json) and getColors() after the last frame (<out>.colors).  A schedule
// entry [mode, frames, "image.ppm"] hands that image to toggleGridMode (modes 1, 2, 7) and writes
// getColors() after that entry's frames to <out>.colors.<entry index>.
const fs = require('fs');
const { MlsMpmSimulator, Conf } = require('./index');

(async () => {
    const [out, maxParticles, particles, frames, schedule] = process.argv.slice(2);
    const conf = new Conf(false);
    conf.maxParticles = Number(maxParticles);
    conf.particles = Number(particles);
    conf.updateParams();
    let now = 50000;
    // FLOW_SLABS='[0,60,128]': the same run as a slab group (one process, all slabs on the one GPU)
    const slabs = process.env.FLOW_SLABS ? { planeStarts: JSON.parse(process.env.FLOW_SLABS) } : null;
    const sim = new MlsMpmSimulator(null, { conf, rebinInterval: 2, clock: () => now, slabs });
    await sim.init();
    fs.writeFileSync(out + '.init', Buffer.from(sim.particleBuffer.floatArray.buffer));
    if (schedule) {
        const packs = [], info = [];
        let f = 0;
        let entry = 0;
        for (const [mode, count, image] of JSON.parse(schedule)) {
            await sim.toggleGridMode(mode, image);

            for (let k = 0; k < count; k++, f++) {
                now += 16;
                await sim.update(1 / 30, 0.5 * f);
                packs.push(Buffer.from(sim.modes.pack()));
                info.push({ dt: sim.uniforms.dt.value, gridMode: sim.uniforms.gridMode.value,
                    frontGravityMode: sim.uniforms.frontGravityMode.value, time: sim.uniforms.time.value });
            }
            if (image) fs.writeFileSync(out + '.colors.' + entry, Buffer.from(sim.getColors().buffer));
            entry++;
        }
        fs.writeFileSync(out + '.modes', Buffer.concat(packs));
        fs.writeFileSync(out + '.frames.json', JSON.stringify(info));
        fs.writeFileSync(out + '.colors', Buffer.from(sim.getColors().buffer));
    } else {
        for (let f = 0; f < Number(frames); f++) await sim.update(1 / 30, 0.5 * f);
    }
    const rec = sim.readAoS();
    fs.writeFileSync(out, Buffer.from(rec.buffer));
    const pos = sim.getPositions();
    fs.writeFileSync(out + '.pos', Buffer.from(pos.buffer));
    if (slabs) fs.writeFileSync(out + '.group.json', JSON.stringify(sim.domain.info()));
})().catch((e) => { console.error(e); process.exit(1); });
