'use strict';
// Host-side record store with the layout rules of the reference's StructuredArray
// (src/mls-mpm/structuredArray.js:25-120): members in declaration order, each aligned to
// its type's alignment (in floats), record padded to a multiple of four floats; one
// ArrayBuffer seen as `floatArray` and `intArray`; `set(index, element, value)` writes the
// host copy.  Device storage is the HIP library's SoA store, so the three.js storage-node
// members (`buffer`, `element`, `get`, `setAtomic`) do not exist here.

// type name -> [floats, alignment in floats, is float]
const TYPE_TABLE = new Map([
    ['int', [1, 1, false]], ['uint', [1, 1, false]], ['float', [1, 1, true]],
    ['vec2', [2, 2, true]], ['ivec2', [2, 2, false]], ['uvec2', [2, 2, false]],
    ['vec3', [3, 4, true]], ['ivec3', [3, 4, false]], ['uvec3', [3, 4, false]],
    ['vec4', [4, 4, true]], ['ivec4', [4, 4, false]], ['uvec4', [4, 4, false]],
    ['mat2', [4, 2, true]], ['mat3', [12, 4, true]], ['mat4', [16, 4, true]],
]);

// src/mls-mpm/mlsMpmSimulator.js:78-86
const PARTICLE_STRUCT = {
    position: { type: 'vec3' }, density: { type: 'float' }, velocity: { type: 'vec3' },
    mass: { type: 'float' }, C: { type: 'mat3' }, direction: { type: 'vec3' }, color: { type: 'vec3' },
};

function roundUp(value, multiple) { return value + ((multiple - (value % multiple)) % multiple); }

class StructuredArray {
    constructor(layout, length, label) {
        this.label = label;
        this.length = length;
        this.layout = {};
        let cursor = 0;
        for (const name of Object.keys(layout)) {
            const spec = typeof layout[name] === 'string' ? { type: layout[name] } : Object.assign({}, layout[name]);
            const entry = TYPE_TABLE.get(spec.type);
            if (!entry) throw new Error("Unknown type '" + spec.type + "'");
            cursor = roundUp(cursor, entry[1]);
            this.layout[name] = Object.assign(spec, { size: entry[0], isFloat: entry[2], offset: cursor });
            cursor += entry[0];
        }
        this.structSize = roundUp(cursor, 4);
        this.floatArray = new Float32Array(this.structSize * length);
        this.intArray = new Int32Array(this.floatArray.buffer);
    }

    set(index, element, value) {
        const member = this.layout[element];
        if (!member) throw new Error("Unknown element '" + element + "'");
        const target = member.isFloat ? this.floatArray : this.intArray;
        const at = index * this.structSize + member.offset;
        if (member.size === 1) {
            if (typeof value !== 'number') throw new TypeError("Expected a Number value for element '" + element + "'");
            target[at] = value;
            return;
        }
        const items = Array.isArray(value) ? value : [value.x, value.y || 0, value.z || 0, value.w || 0];
        if (items.length < member.size) {
            throw new RangeError('Expected an array of length ' + member.size + " for element '" + element + "'");
        }
        for (let k = 0; k < member.size; k++) target[at + k] = items[k];
    }
}

module.exports = { StructuredArray, PARTICLE_STRUCT };
