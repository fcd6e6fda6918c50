#!/bin/bash
# Throughput of the mode scenes against the drift divisor of the riding re-bin: tools/divisor_sweep_modes.sh "<modes>" d1 d2 ...
modes=$1; shift
mkdir -p gpurun_out
for m in $modes; do
for d in "$@"; do
    FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 100 --warmup 20 --mode $m > gpurun_out/mode_div.json 2> gpurun_out/mode_div.err || { echo "$m divisor $d failed"; tail -3 gpurun_out/mode_div.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/mode_div.json"))
print("%-24s divisor %6s: %6.2f G/s %.3f ms | rebins %3d | %s" % ("$m", "$d", d["value"] / 1e9, d["ms_per_step"], d["rebin"]["performed"], {k: x["avg_ms"] for k, x in d["kernels"].items() if k in ("p2g1", "p2g2", "g2p")}))
PY
done
done
