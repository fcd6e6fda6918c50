#!/bin/bash
# Steady-state throughput against the work-item size (FLOWMPM_CHUNK -> flowmpm_set_option): tools/chunk_steady.sh "<bench args>" 4096 2048 ...
args=$1; shift
mkdir -p gpurun_out
for c in "$@"; do
    FLOWMPM_CHUNK=$c timeout -k 10 300 python -u bench.py --no-cpu-baseline $args > gpurun_out/chunk_$c.json 2> gpurun_out/chunk_$c.err || { echo "chunk $c failed"; tail -3 gpurun_out/chunk_$c.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/chunk_$c.json")); s = d["steady"]
print("chunk %5s %-28s first %6.2f steady %6.2f G/s (windows %s) | steady kernels %s" % ("$c", "$args", d["value"] / 1e9, s["value"] / 1e9, " ".join("%.2f" % (w / 1e9) for w in s["windows"]), s["kernel_avg_ms"]))
PY
done
