'use strict';
// Host side of the simulator's display modes: the uniform set-up of toggleGridMode and its
// helpers (src/mls-mpm/mlsMpmSimulator.js, "SIM", lines 1548-2421) and the boost clocks of
// update() (SIM:2426-2494).  Math.random of the reference (SIM:2242-2253, 2298) is replaced
// by the seeded counter generator of particles.js, so this file and the Python mirror
// (flow_amd/modes.py) write the same colours.
//
// toggleGridMode rewrites the colour floats of the host array and binds a NEW storage buffer
// made from it to the render material (SIM:2337-2384); the compute kernels keep the buffer
// node they were built with (structuredArray.js:75-77), so the simulated state and the
// colours g2p reads do not change.  `renderColors` is the colour column of that new buffer.
const { PALETTE, paletteRgb, uniform01 } = require('./particles');
const { Vector3 } = require('./vector3');

const COLOR_SPHERES_BOOST_DURATION = 1.4;      // SIM:34
const SEGMENT_NAMES = ['Pink', 'Purple', 'Cyan', 'Teal', 'DarkBlue', 'DarkPurple', 'DarkRed', 'Orange', 'LightGreen'];
// (sphere angle, negative-space angle) in degrees (SIM:1836, 1878-1879, ..., 2130-2131)
const SEGMENT_ANGLES = [[30, 30], [135, 110], [270, 250], [45, 65], [180, 200], [300, 320], [0, 20], [90, 70], [210, 190]];
const SVG_VERTICES = [[311.096, 125.894], [92.6019, 0], [0.834339, 296.632], [261.155, 235.488],
    [392.251, 199.766], [520.226, 283.002], [558.723, 4.16184]];      // SIM:1660-1668

const uniform = (value) => ({ value });
const lower = (name) => name[0].toLowerCase() + name.slice(1);
const negativeName = (name) => 'negativeSpaceSphereCenter' + (name === 'Pink' ? '' : name);   // SIM:216-224

function polygonVertices() {                    // SIM:1652-1688
    const w = 120, h = 60, x0 = (128 - w) / 2, y0 = (64 - h) / 2;
    return SVG_VERTICES.map(([x, y]) => [x0 + (x / 559) * w, y0 + (y / 302) * h]);
}

function colorSphereLayout() {                  // SIM:1704-1762
    const rgb = paletteRgb();
    const unique = [], seen = new Set();
    PALETTE.forEach((hex, i) => { if (!seen.has(hex)) { seen.add(hex); unique.push(rgb[i]); } });
    const rows = [3, 4, 3], spacing = 21;
    const startY = 32 - (rows.length - 1) * spacing / 2;
    const positions = [];
    rows.forEach((n, row) => {
        const startX = 64 - (n - 1) * spacing / 2;
        for (let col = 0; col < n && positions.length < unique.length; col++)
            positions.push([startX + col * spacing, startY + row * spacing, 32]);
    });
    return { positions, unique };
}

function imageGrid(width, height, numParticles) {   // SIM:1565-1602
    const aspect = width / height;
    let gh = Math.floor(Math.sqrt(numParticles / aspect));
    let gw = Math.floor(gh * aspect);
    while (gw * gh < numParticles) {
        if (gw / gh < aspect) gw++; else gh++;
    }
    const spacing = Math.min(120 / gw, 60 / gh);
    return { gw, gh, startX: (128 - gw * spacing) / 2, startY: (64 - gh * spacing) / 2, spacing };
}

// pixels: {width, height, data} with `channels` bytes per pixel (RGBA like ImageData by default).
function imageColors(pixels, gw, gh, numParticles) {   // SIM:1605-1632
    const ch = pixels.channels || 4, out = [];
    for (let i = 0; i < numParticles; i++) {
        const gx = i % gw, gy = Math.floor(i / gw);
        if (gy >= gh) continue;
        const nx = gw > 1 ? gx / (gw - 1) : 0, ny = gh > 1 ? gy / (gh - 1) : 0;
        const ix = Math.floor((1.0 - nx) * (pixels.width - 1)), iy = Math.floor((1.0 - ny) * (pixels.height - 1));
        const p = (iy * pixels.width + ix) * ch;
        out.push([pixels.data[p] / 255, pixels.data[p + 1] / 255, pixels.data[p + 2] / 255]);
    }
    return out;
}

function closestPalette(color, rgb) {           // SIM:2196-2214
    let best = rgb[0], min = Infinity;
    for (const c of rgb) {
        const d = Math.sqrt((color[0] - c[0]) ** 2 + (color[1] - c[1]) ** 2 + (color[2] - c[2]) ** 2);
        if (d < min) { min = d; best = c; }
    }
    return best;
}

// Binary PPM (P6, maxval 255) -> {width, height, channels: 3, data}.  Stands in for the browser's
// Image + canvas decode (SIM:1550-1559); PNG / JPEG decoding is out of reach here.
function readPpm(path) {
    const data = require('fs').readFileSync(path);
    const tokens = [];
    let pos = 0;
    const isSpace = (c) => c === 32 || (c >= 9 && c <= 13);
    while (tokens.length < 4) {
        while (isSpace(data[pos])) pos++;
        if (data[pos] === 35) { pos = data.indexOf(10, pos) + 1; continue; }     // '#' comment
        let end = pos;
        while (!isSpace(data[end])) end++;
        tokens.push(data.toString('latin1', pos, end));
        pos = end;
    }
    if (tokens[0] !== 'P6' || Number(tokens[3]) !== 255) throw new Error(path + ': only binary PPM (P6) with maxval 255 is read');
    const width = Number(tokens[1]), height = Number(tokens[2]);
    return { width, height, channels: 3, data: new Uint8Array(data.buffer, data.byteOffset + pos + 1, width * height * 3) };
}

// Field order of flowmpm_mode_uniforms (include/flowmpm.h); every field is a 4-byte word.
const LAYOUT = [['cylinder_mode', 'u', 1], ['sphere_containment_mode', 'u', 1], ['polygon_containment_mode', 'u', 1],
    ['color_spheres_mode', 'u', 1], ['two_color_sphere_mode', 'u', 1], ['segment_mode', 'u', 9],
    ['cylinder_radius', 'f', 1], ['cylinder_center_x', 'f', 1], ['cylinder_center_y', 'f', 1],
    ['sphere_radius', 'f', 1], ['sphere_center', 'f', 3], ['polygon_vertex_count', 'u', 1],
    ['polygon_vertices', 'f', 32], ['color_spheres_boost_time', 'f', 1], ['color_spheres_boost_duration', 'f', 1],
    ['color_sphere_radius', 'f', 1], ['color_sphere_positions', 'f', 39], ['palette_colors', 'f', 39],
    ['segment_boost_time', 'f', 9], ['segment_boost_duration', 'f', 9], ['segment_main_radius', 'f', 9],
    ['segment_sphere_radius', 'f', 9], ['segment_sphere_center', 'f', 27], ['segment_negative_center', 'f', 27],
    ['negative_space_sphere_radius', 'f', 1], ['grid_width', 'u', 1], ['grid_height', 'u', 1],
    ['grid_start_x', 'f', 1], ['grid_start_y', 'f', 1], ['grid_spacing_x', 'f', 1], ['grid_spacing_y', 'f', 1],
    ['grid_z', 'f', 1], ['grid_random_offset', 'f', 1], ['grid_noise_strength', 'f', 1]];
const MODE_UNIFORM_WORDS = LAYOUT.reduce((n, f) => n + f[2], 0);

// fields: {name: number | flat array} -> ArrayBuffer in the struct's layout.
function packFields(fields) {
    const buf = new ArrayBuffer(MODE_UNIFORM_WORDS * 4);
    const f32 = new Float32Array(buf), u32 = new Uint32Array(buf);
    let at = 0;
    for (const [name, type, count] of LAYOUT) {
        const v = fields[name];
        const flat = v === undefined ? [] : (Array.isArray(v) ? v.flat(2) : [v]);
        for (let i = 0; i < count && i < flat.length; i++) (type === 'u' ? u32 : f32)[at + i] = flat[i];
        at += count;
    }
    return buf;
}

class ModeHost {
    constructor(uniforms, clock, seed) {
        this.u = uniforms;
        this.clock = clock || (() => Date.now());
        this.seed = seed === undefined ? 0x0C0104ED : seed;
        this.toggles = 0;
        this.gridMode = false;
        this.currentImageMode = null;
        this.gridTargetColors = [];
        this.polygonVertices = [];
        this.colorSpherePositions = [];
        this.colorSpheresModeStartTime = 0;
        this.segmentModeStartTime = new Array(9).fill(0);
        this.renderColors = null;
        const u = uniforms;
        u.cylinderMode = uniform(0); u.cylinderRadius = uniform(44.0);              // SIM:154-161
        u.cylinderCenterX = uniform(64.0); u.cylinderCenterY = uniform(32.0);
        u.sphereContainmentMode = uniform(0); u.polygonContainmentMode = uniform(0);
        u.colorSpheresMode = uniform(0); u.twoColorSphereMode = uniform(0);
        u.polygonVertexCount = uniform(0);                                          // SIM:172-176
        u.polygonVertices = [];
        for (let i = 0; i < 16; i++) u.polygonVertices.push(uniform([0, 0]));
        u.colorSpheresBoostTime = uniform(0);                                       // SIM:177-178
        u.colorSpheresBoostDuration = uniform(COLOR_SPHERES_BOOST_DURATION);
        for (const name of SEGMENT_NAMES) {                                         // SIM:162-170, 179-234
            u['populationSegment' + name + 'Mode'] = uniform(0);
            u['populationSegment' + name + 'BoostTime'] = uniform(0);
            u['populationSegment' + name + 'BoostDuration'] = uniform(COLOR_SPHERES_BOOST_DURATION);
            u['populationSegment' + name + 'MainSphereRadius'] = uniform(18.0);
            u[lower(name) + 'SphereRadius'] = uniform(7.0);
            u[lower(name) + 'SphereCenter'] = uniform(new Vector3());
            u[negativeName(name)] = uniform(new Vector3());
        }
        u.negativeSpaceSphereRadius = uniform(17.0);                                // SIM:215
        u.sphereRadius = uniform(25.0); u.colorSphereRadius = uniform(7.0);         // SIM:233, 235
        u.sphereCenterX = uniform(64.0); u.sphereCenterY = uniform(32.0); u.sphereCenterZ = uniform(32.0);
        u.colorSpherePositions = []; u.paletteColors = [];                          // SIM:240-245
        for (let i = 0; i < 13; i++) {
            u.colorSpherePositions.push(uniform(new Vector3()));
            u.paletteColors.push(uniform(new Vector3()));
        }
        u.gridWidth = uniform(0); u.gridHeight = uniform(0);                        // SIM:246-254
        u.gridStartX = uniform(0); u.gridStartY = uniform(0); u.gridSpacingX = uniform(0); u.gridSpacingY = uniform(0);
        u.gridZ = uniform(32); u.gridRandomOffset = uniform(0.5); u.gridNoiseStrength = uniform(3.5);
    }

    setupPolygonContainment() {                 // SIM:1652-1702
        this.polygonVertices = polygonVertices();
        this.u.polygonVertexCount.value = this.polygonVertices.length;
        for (let i = 0; i < 16; i++)
            this.u.polygonVertices[i].value = i < this.polygonVertices.length ? this.polygonVertices[i].slice() : [0, 0];
    }

    setupColorSpheres() {                       // SIM:1704-1774
        const { positions, unique } = colorSphereLayout();
        this.colorSpherePositions = positions;
        for (let i = 0; i < 13; i++) {
            if (i < positions.length) this.u.colorSpherePositions[i].value.set(...positions[i]);
            this.u.paletteColors[i].value.set(...unique[Math.min(i, unique.length - 1)]);
        }
    }

    loadImageAndSetupGrid(pixels, numParticles) {   // SIM:1548-1650, decoded pixels handed in
        const g = imageGrid(pixels.width, pixels.height, numParticles);
        this.gridTargetColors = imageColors(pixels, g.gw, g.gh, numParticles);
        const u = this.u;
        u.gridWidth.value = g.gw; u.gridHeight.value = g.gh;
        u.gridStartX.value = g.startX; u.gridStartY.value = g.startY;
        u.gridSpacingX.value = g.spacing; u.gridSpacingY.value = g.spacing;
    }

    // SIM:1776-2421.  floatArray / structSize / colorOffset: the host particle array.
    toggle(mode, floatArray, structSize, colorOffset, numParticles, image) {
        const u = this.u;
        const enabled = (mode > 0 && mode < 3) || mode === 7;
        this.gridMode = enabled;
        u.gridMode.value = enabled ? 1 : 0;
        u.frontGravityMode.value = (mode === 3 || mode === 4) ? 1 : 0;
        u.cylinderMode.value = mode === 4 ? 1 : 0;
        u.sphereContainmentMode.value = mode === 5 ? 1 : 0;
        u.polygonContainmentMode.value = mode === 8 ? 1 : 0;
        u.colorSpheresMode.value = mode === 6 ? 1 : 0;
        u.twoColorSphereMode.value = mode === 18 ? 1 : 0;
        if (mode === 8 && this.polygonVertices.length === 0) this.setupPolygonContainment();
        if (mode === 6 && this.colorSpherePositions.length === 0) this.setupColorSpheres();
        if (mode === 6) {                                                           // SIM:1826-1831
            this.colorSpheresModeStartTime = this.clock();
            u.colorSpheresBoostTime.value = COLOR_SPHERES_BOOST_DURATION;
        } else {
            u.colorSpheresBoostTime.value = 0.0;
        }
        SEGMENT_NAMES.forEach((name, k) => {                                        // SIM:1833-2156
            const on = mode === 9 + k;
            u['populationSegment' + name + 'Mode'].value = on ? 1 : 0;
            const boost = u['populationSegment' + name + 'BoostTime'];
            if (!on) { boost.value = 0.0; return; }
            const cx = u.sphereCenterX.value, cy = u.sphereCenterY.value, cz = u.sphereCenterZ.value;
            const sphereRad = (SEGMENT_ANGLES[k][0] * Math.PI) / 180.0, chunkRad = (SEGMENT_ANGLES[k][1] * Math.PI) / 180.0;
            const dist = 40.0, near = dist * 0.5;
            u[lower(name) + 'SphereCenter'].value.set(cx + dist * Math.cos(sphereRad), cy + dist * Math.sin(sphereRad), cz);
            u[negativeName(name)].value.set(cx + near * Math.cos(chunkRad), cy + near * Math.sin(chunkRad), cz);
            this.segmentModeStartTime[k] = this.clock();
            boost.value = COLOR_SPHERES_BOOST_DURATION;
        });

        const n = numParticles, rgb = paletteRgb();
        const draw0 = 8 * this.toggles;
        this.toggles += 1;
        const write = (i, c) => {
            const at = i * structSize + colorOffset;
            floatArray[at] = c[0]; floatArray[at + 1] = c[1]; floatArray[at + 2] = c[2];
        };
        if (enabled) {                                                              // SIM:2163-2277
            if (this.currentImageMode !== mode) {
                if (!image) throw new Error('mode ' + mode + ' needs the decoded image {width, height, data} or a PPM path');
                this.loadImageAndSetupGrid(typeof image === 'string' ? readPpm(image) : image, n);
                this.currentImageMode = mode;
            }
            const target = this.gridTargetColors;
            for (let i = 0; i < n && i < target.length; i++) {
                let color = target[i];
                if (mode === 7) {
                    color = closestPalette(color, rgb);
                } else if (uniform01(this.seed, i, draw0) < 0.6) {
                    const near = closestPalette(color, rgb);
                    const u1 = Math.max(uniform01(this.seed, i, draw0 + 1), Math.pow(2, -33));
                    const u2 = uniform01(this.seed, i, draw0 + 2);
                    const z0 = Math.sqrt(-2.0 * Math.log(u1)) * Math.cos(2.0 * Math.PI * u2);
                    const f = Math.max(0, Math.min(1, Math.abs(z0 * 0.25 + 0.1))), g = 1.0 - f;
                    color = [near[0] * f + color[0] * g, near[1] * f + color[1] * g, near[2] * f + color[2] * g];
                }
                write(i, color);
            }
        } else {
            if (mode !== 3) this.currentImageMode = null;                           // SIM:2280-2282
            if (mode === 18) {                                                      // SIM:2285-2310
                const c1 = [0x31 / 255, 0x72 / 255, 0x85 / 255], c2 = [0x79 / 255, 0xC8 / 255, 0xD6 / 255];
                for (let i = 0; i < n; i++) write(i, uniform01(this.seed, i, draw0) < 0.5 ? c1 : c2);
            } else {                                                                // SIM:2312-2334
                for (let i = 0; i < n; i++) write(i, rgb[i % rgb.length]);
            }
        }
        const count = floatArray.length / structSize;                               // SIM:2337-2341
        this.renderColors = new Float32Array(count * 3);
        for (let i = 0; i < count; i++)
            for (let a = 0; a < 3; a++) this.renderColors[3 * i + a] = floatArray[i * structSize + colorOffset + a];
    }

    updateBoostClocks() {                       // SIM:2426-2494
        const u = this.u;
        if (u.colorSpheresMode.value === 1 && this.colorSpheresModeStartTime > 0) {
            const since = (this.clock() - this.colorSpheresModeStartTime) / 1000;
            u.colorSpheresBoostTime.value = Math.max(0, COLOR_SPHERES_BOOST_DURATION - since);
        }
        SEGMENT_NAMES.forEach((name, k) => {
            if (u['populationSegment' + name + 'Mode'].value === 1 && this.segmentModeStartTime[k] > 0) {
                const since = (this.clock() - this.segmentModeStartTime[k]) / 1000;
                u['populationSegment' + name + 'BoostTime'].value = Math.max(0, COLOR_SPHERES_BOOST_DURATION - since);
            }
        });
    }

    // The current uniform values as a flowmpm_mode_uniforms record (ArrayBuffer).
    pack() {
        const u = this.u;
        const seg = (fmt) => SEGMENT_NAMES.map((name) => {
            const v = u[fmt(name)].value;
            return v instanceof Vector3 ? v.toArray() : v;
        });
        return packFields({
            cylinder_mode: u.cylinderMode.value, sphere_containment_mode: u.sphereContainmentMode.value,
            polygon_containment_mode: u.polygonContainmentMode.value, color_spheres_mode: u.colorSpheresMode.value,
            two_color_sphere_mode: u.twoColorSphereMode.value,
            segment_mode: seg((n) => 'populationSegment' + n + 'Mode'),
            cylinder_radius: u.cylinderRadius.value, cylinder_center_x: u.cylinderCenterX.value,
            cylinder_center_y: u.cylinderCenterY.value, sphere_radius: u.sphereRadius.value,
            sphere_center: [u.sphereCenterX.value, u.sphereCenterY.value, u.sphereCenterZ.value],
            polygon_vertex_count: u.polygonVertexCount.value,
            polygon_vertices: u.polygonVertices.map((p) => p.value),
            color_spheres_boost_time: u.colorSpheresBoostTime.value,
            color_spheres_boost_duration: u.colorSpheresBoostDuration.value,
            color_sphere_radius: u.colorSphereRadius.value,
            color_sphere_positions: u.colorSpherePositions.map((p) => p.value.toArray()),
            palette_colors: u.paletteColors.map((p) => p.value.toArray()),
            segment_boost_time: seg((n) => 'populationSegment' + n + 'BoostTime'),
            segment_boost_duration: seg((n) => 'populationSegment' + n + 'BoostDuration'),
            segment_main_radius: seg((n) => 'populationSegment' + n + 'MainSphereRadius'),
            segment_sphere_radius: seg((n) => lower(n) + 'SphereRadius'),
            segment_sphere_center: seg((n) => lower(n) + 'SphereCenter'),
            segment_negative_center: seg(negativeName),
            negative_space_sphere_radius: u.negativeSpaceSphereRadius.value,
            grid_width: u.gridWidth.value, grid_height: u.gridHeight.value,
            grid_start_x: u.gridStartX.value, grid_start_y: u.gridStartY.value,
            grid_spacing_x: u.gridSpacingX.value, grid_spacing_y: u.gridSpacingY.value,
            grid_z: u.gridZ.value, grid_random_offset: u.gridRandomOffset.value,
            grid_noise_strength: u.gridNoiseStrength.value,
        });
    }
}

module.exports = { ModeHost, readPpm, packFields, polygonVertices, colorSphereLayout, imageGrid, imageColors,
    closestPalette, COLOR_SPHERES_BOOST_DURATION, SEGMENT_NAMES, SEGMENT_ANGLES, MODE_UNIFORM_WORDS };
