"""The C ABI from a plain C program (tests/c_abi_driver.c, built by __graft_entry__.build()), and the
struct layouts every binding re-declares against the header's."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest

from flow_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "bin", "c_abi_driver")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _driver():
    assert os.path.exists(DRIVER), "tests/bin/c_abi_driver is missing: run __graft_entry__.build()"
    return DRIVER


def test_struct_sizes_agree_between_c_ctypes_and_javascript():
    """sizeof as gcc lays the header's structs out (C, not C++), as the library was compiled, as
    flow_amd/_lib.py re-declares them, and the word count host_js/modes.js packs."""
    r = subprocess.run([_driver(), "--sizes"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    header, library = ([int(x) for x in line.split()[1:]] for line in r.stdout.strip().splitlines())
    assert header == library
    mine = [ctypes.sizeof(t) for t in (_lib.Config, _lib.Uniforms, _lib.ModeUniforms, _lib.Timings)]
    assert mine == library
    sizes = (ctypes.c_int64 * 4)()
    assert _lib.load().flowmpm_abi_sizes(sizes) == 0 and list(sizes) == library
    # host_js/modes.js packs flowmpm_mode_uniforms word by word from its LAYOUT table
    js = open(os.path.join(ROOT, "host_js", "modes.js")).read()
    layout = js[js.index("const LAYOUT"):js.index("const MODE_UNIFORM_WORDS")]
    table = [(m.group(1), int(m.group(2))) for m in re.finditer(r"\[\s*'(\w+)'\s*,\s*'[uf]'\s*,\s*(\d+)\s*\]", layout)]
    assert [(name, 4 * words) for name, words in table] == \
        [(name, getattr(_lib.ModeUniforms, name).size) for name, _ in _lib.ModeUniforms._fields_]
    assert 4 * sum(words for _, words in table) == library[2]


def test_field_offsets_of_the_ctypes_structs_follow_the_header():
    """Offsets, not only sizes: every field of the two uniform structs in the order and at the offset
    the header gives it (parsed from include/flowmpm.h: all members are 4-byte scalars or arrays)."""
    text = open(os.path.join(ROOT, "include", "flowmpm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(\d+)\b", text)}
    structs = {m.group(2): m.group(1) for m in re.finditer(r"typedef struct \{([^}]*)\}\s*(\w+);", text)}
    for c_name, ct in (("flowmpm_uniforms", _lib.Uniforms), ("flowmpm_mode_uniforms", _lib.ModeUniforms)):
        body = structs[c_name]
        offset, fields = 0, []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            base, rest = decl.split(None, 1)
            assert base in ("float", "uint32_t", "int32_t"), decl
            for item in rest.split(","):
                m = re.match(r"\s*(\w+)((?:\[\w+\])*)\s*$", item)
                count = 1
                for dim in re.findall(r"\[(\w+)\]", m.group(2)):
                    count *= int(dim) if dim.isdigit() else consts[dim]
                fields.append((m.group(1), offset, 4 * count))
                offset += 4 * count
        got = [(name, getattr(ct, name).offset, getattr(ct, name).size) for name, _ in ct._fields_]
        assert got == fields, c_name
        assert offset == ctypes.sizeof(ct)


@pytest.mark.gpu
def test_c_driver_runs_the_golden_case_through_the_five_kernels(tmp_path):
    d = np.load(os.path.join(GOLDEN, "small_16cube_512.npz"))
    for name in ("particles_in", "cell_after_p2g1", "cell_after_p2g2", "cellf_after_update", "particles_step1",
                 "particles_step4", "cell_step4"):
        np.ascontiguousarray(d[name]).tofile(os.path.join(tmp_path, name + ".bin"))
    meta = json.load(open(os.path.join(GOLDEN, "small_16cube_512.json")))
    assert meta["grid"] == [16, 16, 16] and meta["particles"] == 512 and meta["steps_after_first"] == 3
    u = meta["uniforms"]
    values = [u["dt"], u["stiffness"], u["rest_density"], u["dynamic_viscosity"], u["noise"], u["time"],
              *u["mouse_ray_direction"], *u["mouse_ray_origin"], *u["mouse_force"], u["cursor_interaction"]]
    r = subprocess.run([_driver(), str(tmp_path)] + [repr(float(x)) for x in values], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok   ") == 12 and "0 failure(s)" in r.stdout, r.stdout
