#!/bin/bash
# Builds library variants with different -D flags (on the CPU box) for A/B runs:
#   tools/variants.sh name1 "-DFOO=1" name2 "-DFOO=2" ...
# Each lands in build_variants/<name>.so; run with FLOWMPM_LIBRARY=build_variants/<name>.so.
# Prints VGPRs / scratch of the three hot tiled kernels per variant.
cd "$(dirname "$0")/.."
mkdir -p build_variants
build_one() {
    name=$1; flags=$2
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine -fPIC -shared -Wno-unused-function \
        -Rpass-analysis=kernel-resource-usage $flags -o build_variants/$name.so flow_amd/csrc/flowmpm.hip > build_variants/$name.log 2>&1
    if grep -q "error" build_variants/$name.log; then echo "$name: BUILD FAILED"; grep error build_variants/$name.log | head -3; return; fi
    python3 - "$name" <<'PY'
import re, sys
name = sys.argv[1]
txt = open("build_variants/%s.log" % name).read()
out = []
for k in ("g2p_tiled", "p2g_momentum_tiled", "p2g_mass_tiled"):
    m = re.search(r"Function Name: \S*%s\S*.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)" % k, txt, re.S)
    if m: out.append("%s vgpr=%s scratch=%s occ=%s" % (k, m.group(1), m.group(2), m.group(3)))
print(name + ": " + "; ".join(out))
PY
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
