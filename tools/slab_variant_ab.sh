#!/bin/bash
# The eight-slab step on one GPU (tools/slab_group_bench.py) for library variants under build_variants/, with the
# kernel statistics of each: tools/slab_variant_ab.sh name1 name2 ...
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
    FLOWMPM_LIBRARY=$PWD/build_variants/$v.so timeout -k 10 300 python3 -u tools/slab_group_bench.py --world 8 --steps 40 --windows 2 > gpurun_out/slab_ab_$v.json 2> gpurun_out/slab_ab_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/slab_ab_$v.err; exit 1; }
    python3 - <<PY
import json
d = json.load(open("gpurun_out/slab_ab_$v.json"))
print("%-10s ms per slab step %s | re-bins per slab %s" % ("$v", [w["ms_per_slab_step"] for w in d["windows"]], d["rebins_per_slab"]))
PY
done
