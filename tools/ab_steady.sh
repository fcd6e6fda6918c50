#!/bin/bash
# tools/ab.sh with the steady windows: tools/ab_steady.sh "<bench args>" name1 name2 ...
args=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
    FLOWMPM_LIBRARY=$PWD/build_variants/$v.so timeout -k 10 300 python -u bench.py --no-cpu-baseline $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_$v.json")); s = d["steady"]
print("%-8s %-24s first %6.2f steady %6.2f G/s | rebins %3d / %3d | stragglers/step %8.0f" % ("$v", "$args", d["value"] / 1e9, s["value"] / 1e9, d["rebin"]["performed"], s["rebins"], s["stragglers_per_step"]))
PY
done
