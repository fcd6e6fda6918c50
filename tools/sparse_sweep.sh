#!/bin/bash
# Below how many particles a tile gets no work item (FLOWMPM_SPARSE_TILE): first window / steady windows.
TAG=${1:-sparse}
mkdir -p gpurun_out/$TAG
for cfg in "--noise 1.0" "" "--mode sphere"; do
for c in 0 8 32 64 128 256; do
  FLOWMPM_SPARSE_TILE=$c timeout -k 10 300 python -u bench.py --no-cpu-baseline $cfg > gpurun_out/$TAG/tmp.json || exit 1
  python - "$cfg" $c gpurun_out/$TAG/tmp.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3])); s = d["steady"]
print("%-16s sparse %4s: first %6.2f steady %6.2f G/s | stragglers/step %8.0f | %s" % (sys.argv[1], sys.argv[2], d["value"] / 1e9, s["value"] / 1e9, s["stragglers_per_step"], s["kernel_avg_ms"]))
PY
done; done
