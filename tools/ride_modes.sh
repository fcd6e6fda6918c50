#!/bin/bash
# Steady state with the three re-bin forms: 0 standalone, 1 riding with its own counting launch, 2 riding with
# the count in the mass pass.  usage: tools/ride_modes.sh <tag> [bench args]
TAG=${1:-ridemodes}; shift
mkdir -p gpurun_out/$TAG
for noise in 0.0 1.0; do for m in 0 1 2; do
  FLOWMPM_RIDE_REBINS=$m timeout -k 10 300 python -u bench.py --no-cpu-baseline --noise $noise "$@" > gpurun_out/$TAG/ride${m}_n$noise.json || exit 1
  python - gpurun_out/$TAG/ride${m}_n$noise.json $m $noise <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); s = d["steady"]
print("noise %s re-bin form %s: first %6.2f steady %6.2f min %6.2f G/s | rebins %3d stragglers/step %7.0f | %s" % (
    sys.argv[3], sys.argv[2], d["value"] / 1e9, s["value"] / 1e9, s["min"] / 1e9, s["rebins"], s["stragglers_per_step"], s["kernel_avg_ms"]))
PY
done; done
