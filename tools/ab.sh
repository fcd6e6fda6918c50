#!/bin/bash
# Runs bench.py once per library variant: tools/ab.sh "<bench args>" name1 name2 ...
args=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
    FLOWMPM_LIBRARY=$PWD/build_variants/$v.so timeout -k 10 300 python -u bench.py --no-cpu-baseline $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_$v.json"))
print("%-10s %6.2f Gps %6.3f ms " % ("$v", d["value"] / 1e9, d["ms_per_step"]),
      {k: x["avg_ms"] for k, x in d["kernels"].items()}, "rebins", d["rebin"]["performed"], "ms each %.3f" % (d["rebin"]["scan_ms"] / max(d["rebin"]["performed"], 1)))
PY
done
