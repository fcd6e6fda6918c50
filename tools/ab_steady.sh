#!/bin/bash
# tools/ab.sh with the steady windows: tools/ab_steady.sh "<bench args>" name1 name2 ...
args=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
    FLOWMPM_LIBRARY=$PWD/build_variants/$v.so timeout -k 10 300 python -u bench.py --no-cpu-baseline $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_$v.json")); s = d["steady"]
short = lambda k: " ".join("%s %.4f" % (n, k[n]) for n in ("p2g1", "p2g2", "updateGrid", "g2p") if n in k)
print("%-8s %-24s first %6.2f steady %6.2f G/s | rebins %3d / %3d | stragglers/step %6.0f | first: %s | steady: %s" % ("$v", "$args", d["value"] / 1e9, s["value"] / 1e9, d["rebin"]["performed"], s["rebins"], s["stragglers_per_step"], short({k: x["avg_ms"] for k, x in d["kernels"].items()}), short(s["kernel_avg_ms"])))
PY
done
