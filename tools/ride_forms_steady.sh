#!/bin/bash
# Steady-window kernel times with the riding re-bin (2), the counting launch of its own (1) and the standalone re-bin (0).
mkdir -p gpurun_out
for r in 2 0 1; do
    FLOWMPM_RIDE_REBINS=$r timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/ride_$r.json 2> gpurun_out/ride_$r.err || { echo "ride $r failed"; tail -3 gpurun_out/ride_$r.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ride_$r.json")); s = d["steady"]
print("ride %s first %6.2f steady %6.2f G/s rebins %d | first-window kernels %s | steady kernels %s" % ("$r", d["value"] / 1e9, s["value"] / 1e9, s["rebins"], {k: v["avg_ms"] for k, v in d["kernels"].items() if k in ("p2g1", "p2g2", "g2p")}, s["kernel_avg_ms"]))
PY
done
