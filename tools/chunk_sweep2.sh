#!/bin/bash
# Upper bound of the particles per work item (FLOWMPM_CHUNK) with tiles cut into equal parts.
TAG=${1:-chunk2}
mkdir -p gpurun_out/$TAG
for cfg in "" "--noise 1.0" "--particles-log2 25 --scene dam_break --steps 100 --steady-steps 50"; do
for c in 1536 2048 3072 4096; do
  FLOWMPM_CHUNK=$c timeout -k 10 300 python -u bench.py --no-cpu-baseline $cfg > gpurun_out/$TAG/tmp.json || exit 1
  python - "$cfg" $c gpurun_out/$TAG/tmp.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3])); s = d["steady"]
print("%-60s chunk %5s: first %6.2f steady %6.2f G/s | %s" % (sys.argv[1], sys.argv[2], d["value"] / 1e9, s["value"] / 1e9, s["kernel_avg_ms"]))
PY
done; done
