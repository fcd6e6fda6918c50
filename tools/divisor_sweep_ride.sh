#!/bin/bash
# Steady-state throughput against the drift divisor (re-bin when particles outside their arena, summed
# over the steps, exceed n / divisor), with the riding re-bin.  usage: tools/divisor_sweep_ride.sh <tag> [bench args]
TAG=${1:-div}; shift
mkdir -p gpurun_out/$TAG
for noise in 0.0 1.0; do
for div in ${DIVS:-1024 2048 4096 8192 16384}; do
  FLOWMPM_DRIFT_DIVISOR=$div timeout -k 10 300 python -u bench.py --no-cpu-baseline --noise $noise "$@" > gpurun_out/$TAG/div_${div}_n$noise.json || exit 1
  python - gpurun_out/$TAG/div_${div}_n$noise.json $div $noise <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); s = d["steady"]
print("noise %s divisor %5s: first %6.2f steady %6.2f min %6.2f G/s | rebins %3d %.3f ms/step stragglers/step %8.0f | %s" % (
    sys.argv[3], sys.argv[2], d["value"] / 1e9, s["value"] / 1e9, s["min"] / 1e9, s["rebins"], s["rebin_ms_per_step"], s["stragglers_per_step"], s["kernel_avg_ms"]))
PY
done; done
