#!/bin/bash
# Time per slab step of the 8-slab decomposition on one GPU (tools/slab_group_bench.py) for several drift divisors of
# the re-bin rule: tools/divisor_sweep_slab.sh d1 d2 ...
mkdir -p gpurun_out
for d in "$@"; do
    FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 300 python3 -u tools/slab_group_bench.py --world 8 --steps 40 --windows 2 > gpurun_out/slab_div_$d.json 2> gpurun_out/slab_div_$d.err || { echo "divisor $d failed"; tail -3 gpurun_out/slab_div_$d.err; exit 1; }
    python3 - <<PY
import json
d = json.load(open("gpurun_out/slab_div_$d.json"))
print("divisor %6s: ms per slab step %s | re-bins per slab %s | stragglers per step %s" % ("$d", [w["ms_per_slab_step"] for w in d["windows"]], d["rebins_per_slab"], [int(x) for x in d["stragglers_per_step_per_slab"]]))
PY
done
