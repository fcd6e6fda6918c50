"""Host-side particle record store with the reference's struct layout.

Mirrors the surface of the reference's StructuredArray
(src/mls-mpm/structuredArray.js:25-120): a `{name: {type}}` layout is parsed into
float offsets with WGSL-like alignment, one buffer is exposed as `floatArray`
(float32) and `intArray` (int32 alias), `set(index, element, value)` writes the host
array, and `structSize` is the record length in floats.  The three.js storage node
(`buffer`, `element`, `get`, `setAtomic`) has no counterpart: device storage lives in
the HIP library as SoA.
"""
import numpy as np

# name -> (size in floats, alignment in floats, is_float)   (structuredArray.js:3-23)
_TYPES = {
    "int": (1, 1, False), "uint": (1, 1, False), "float": (1, 1, True),
    "vec2": (2, 2, True), "ivec2": (2, 2, False), "uvec2": (2, 2, False),
    "vec3": (3, 4, True), "ivec3": (3, 4, False), "uvec3": (3, 4, False),
    "vec4": (4, 4, True), "ivec4": (4, 4, False), "uvec4": (4, 4, False),
    "mat2": (4, 2, True), "mat3": (12, 4, True), "mat4": (16, 4, True),
}

# The particle struct of the simulator (src/mls-mpm/mlsMpmSimulator.js:78-86).
PARTICLE_STRUCT = {
    "position": {"type": "vec3"},
    "density": {"type": "float"},
    "velocity": {"type": "vec3"},
    "mass": {"type": "float"},
    "C": {"type": "mat3"},
    "direction": {"type": "vec3"},
    "color": {"type": "vec3"},
}


class StructuredArray:
    def __init__(self, layout, length, label=None):
        self.layout = self._parse(layout)
        self.length = int(length)
        self.label = label
        self.floatArray = np.zeros(self.structSize * self.length, dtype=np.float32)
        self.intArray = self.floatArray.view(np.int32)

    def _parse(self, layout):
        """structuredArray.js:83-119: members are laid out in declaration order, each
        aligned to its type's alignment; the record is padded to a multiple of 4 floats."""
        offset = 0
        parsed = {}
        for key, member in layout.items():
            if isinstance(member, str):
                member = {"type": member}
            member = dict(member)
            if member["type"] not in _TYPES:
                raise ValueError("Unknown type '%s'" % member["type"])
            size, alignment, is_float = _TYPES[member["type"]]
            offset += (-offset) % alignment
            member.update(size=size, isFloat=is_float, offset=offset)
            offset += size
            parsed[key] = member
        offset += (-offset) % 4
        self.structSize = offset
        return parsed

    def set(self, index, element, value):
        """structuredArray.js:47-73.  The reference logs and returns on bad input;
        here bad input raises."""
        member = self.layout.get(element)
        if member is None:
            raise KeyError("Unknown element '%s'" % element)
        offset = index * self.structSize + member["offset"]
        array = self.floatArray if member["isFloat"] else self.intArray
        if member["size"] == 1:
            if not np.isscalar(value):
                raise TypeError("Expected a number for element '%s'" % element)
            array[offset] = value
            return
        if hasattr(value, "x"):
            value = [value.x, getattr(value, "y", 0), getattr(value, "z", 0), getattr(value, "w", 0)]
        value = list(value)
        if len(value) < member["size"]:
            raise ValueError("Expected an array of length %d for element '%s'" % (member["size"], element))
        array[offset:offset + member["size"]] = value[:member["size"]]

    def records(self):
        """The buffer as a [length, structSize] float32 view."""
        return self.floatArray.reshape(self.length, self.structSize)
