#!/usr/bin/env python3
"""Instruction mix of the hot tiled kernels from a hipcc -save-temps assembly file.

    hipcc ... -save-temps -o lib.so flow_amd/csrc/flowmpm.hip   (in a scratch directory)
    tools/isa_stats.py flowmpm-hip-amdgcn-amd-amdhsa-gfx950.s

Prints, per kernel: instruction count, 64-bit address arithmetic (v_lshl_add_u64 /
v_mad_u64_u32), global accesses with a scalar base (saddr) and with a 64-bit vector address,
LDS operations, and VGPR count.
"""
import re
import sys

KERNELS = ("g2p_tiled_kernelILb0", "g2p_tiled_kernelILb1", "p2g_momentum_tiled", "p2g_mass_tiled")


def main(path):
    text = open(path).read()
    for key in KERNELS:
        m = re.search(r"^(_Z\S*%s\S*):.*?\n(.*?)s_endpgm" % key, text, re.S | re.M)
        if not m:
            continue
        body = m.group(2)
        glob = re.findall(r"^\s*(global_(?:load|store|atomic)\S*)\s+(.*)$", body, re.M)
        saddr = sum(1 for _, ops in glob if re.search(r"s\[\d+:\d+\]", ops))
        vg = re.search(r"\.vgpr_count:\s+(\d+)", text[text.find(".name:           " + m.group(1)):])
        print("%-24s insts=%5d u64_shift_add=%3d u64_mad=%3d global_saddr=%3d global_vaddr=%3d lds=%3d vgpr=%s" % (
            key, len(re.findall(r"^\s+[sv]_|^\s+(?:ds|global|buffer)_", body, re.M)),
            body.count("v_lshl_add_u64"), body.count("v_mad_u64_u32"), saddr, len(glob) - saddr,
            len(re.findall(r"^\s+ds_", body, re.M)), vg.group(1) if vg else "?"))


if __name__ == "__main__":
    main(sys.argv[1])
