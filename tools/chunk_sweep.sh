#!/bin/bash
# Work items per compute unit (FLOWMPM_ITEMS_PER_CU -> particles per work item) on the small configurations.
TAG=${1:-chunk}; export TAG_=$TAG
mkdir -p gpurun_out/$TAG
for cfg in "--particles-log2 20 --grid 128,128,128" "--particles 81920 --grid 128,64,64 --noise 1.0 --steps 400 --warmup 50" "--particles-log2 22 --grid 128,128,128"; do
for per in 1 2 4 8 16 32; do
  FLOWMPM_ITEMS_PER_CU=$per timeout -k 10 300 python -u bench.py --no-cpu-baseline --steady-windows 0 $cfg > gpurun_out/$TAG/tmp.json || exit 1
  python - "$cfg" $per <<'PY'
import json, sys
d = json.load(open("gpurun_out/%s/tmp.json" % "TAGX".replace("TAGX", __import__("os").environ.get("TAG_", "chunk"))))
print("%-70s items/CU %2s: %6.2f G/s %7.4f ms | %s" % (sys.argv[1], sys.argv[2], d["value"] / 1e9, d["ms_per_step"], {n: x["avg_ms"] for n, x in d["kernels"].items() if n in ("p2g1", "p2g2", "g2p")}))
PY
done; done
