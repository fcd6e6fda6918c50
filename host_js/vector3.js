'use strict';
// The handful of three.js Vector3 operations the simulator's host code relies on
// (setMouseRay mutates its arguments: src/mls-mpm/mlsMpmSimulator.js:1540-1545).
class Vector3 {
    constructor(x, y, z) { this.x = x || 0; this.y = y || 0; this.z = z || 0; }
    set(x, y, z) { this.x = x; this.y = y; this.z = z; return this; }
    copy(v) { return this.set(v.x, v.y, v.z); }
    clone() { return new Vector3(this.x, this.y, this.z); }
    add(v) { return this.set(this.x + v.x, this.y + v.y, this.z + v.z); }
    sub(v) { return this.set(this.x - v.x, this.y - v.y, this.z - v.z); }
    multiplyScalar(s) { return this.set(this.x * s, this.y * s, this.z * s); }
    divideScalar(s) { return this.multiplyScalar(1 / s); }
    length() { return Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z); }
    normalize() { return this.divideScalar(this.length() || 1); }
    toArray() { return [this.x, this.y, this.z]; }
}
module.exports = { Vector3 };
