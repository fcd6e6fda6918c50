'use strict';
// Initial particle records: the host loop of init() (src/mls-mpm/mlsMpmSimulator.js:89-129)
// with a seeded counter-based generator in place of Math.random, bit-compatible with
// flow_amd/particles.py so both hosts start from identical particles.
const PALETTE = ['#FF56B7', '#BC96F9', '#C5F0F2', '#007487', '#007487', '#153943', '#232356',
    '#50121A', '#FF6425', '#D6F499', '#5CCAD8', '#5CCAD8', '#C5F0F2'];

function hash32(x) {
    x = x >>> 0;
    x ^= x >>> 16; x = Math.imul(x, 0x7FEB352D) >>> 0;
    x ^= x >>> 15; x = Math.imul(x, 0x846CA68B) >>> 0;
    x ^= x >>> 16;
    return x >>> 0;
}

function uniform01(seed, index, draw) {
    let h = hash32((index ^ seed) >>> 0);
    h = hash32((h + (Math.imul(draw, 0x9E3779B9) >>> 0)) >>> 0);
    return h / 4294967296;
}

function paletteRgb() {
    return PALETTE.map((hex) => [parseInt(hex.slice(1, 3), 16) / 255, parseInt(hex.slice(3, 5), 16) / 255,
        parseInt(hex.slice(5, 7), 16) / 255]);
}

// Fills `store` (a StructuredArray of the particle struct) for ids [0, count).
function referenceScene(store, count, grid, seed) {
    const colors = paletteRgb();
    seed = seed >>> 0;
    const massSeed = (seed ^ 0x5BD1E995) >>> 0;
    for (let i = 0; i < count; i++) {
        let x, y, z, attempt = 0;
        for (;;) {                                   // rejection-sample the unit ball (SIM:116-121)
            x = uniform01(seed, i, 3 * attempt) * 2.0 - 1.0;
            y = uniform01(seed, i, 3 * attempt + 1) * 2.0 - 1.0;
            z = uniform01(seed, i, 3 * attempt + 2) * 2.0 - 1.0;
            if (Math.sqrt(x * x + y * y + z * z) <= 1.0) break;
            attempt++;
        }
        const pos = [(x * 0.8 + 1.0) / 2.0 * grid[0], (y * 0.8 + 1.0) / 2.0 * grid[1], (z * 0.8 + 1.0) / 2.0 * grid[2]];
        store.set(i, 'position', pos);
        store.set(i, 'mass', 1.0 - uniform01(massSeed, i, 0) * 0.002);
        store.set(i, 'color', colors[i % colors.length]);
    }
}

module.exports = { PALETTE, hash32, uniform01, paletteRgb, referenceScene };
