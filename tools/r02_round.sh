#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r02_round.sh <tag>
# The bench lines of the round: default workload, the reference's default noise, configuration 2,
# configuration 1's real size, and the sustained-clock stamps (diagnostic build, if present).
set -e
TAG=${1:-r02a}
mkdir -p gpurun_out/$TAG
B="timeout -k 10 300 python -u bench.py --no-cpu-baseline"
$B > gpurun_out/$TAG/bench_16M_256cube.json 2> gpurun_out/$TAG/bench.err
$B --noise 1.0 > gpurun_out/$TAG/bench_16M_256cube_noise1.json 2>> gpurun_out/$TAG/bench.err
$B --particles-log2 20 --grid 128,128,128 > gpurun_out/$TAG/bench_cfg2_1M_128cube.json 2>> gpurun_out/$TAG/bench.err
$B --particles 81920 --grid 128,64,64 --noise 1.0 --steps 400 --warmup 50 > gpurun_out/$TAG/bench_cfg1_81920_128x64x64.json 2>> gpurun_out/$TAG/bench.err
if [ -f build_variants/stamps.so ]; then
    FLOWMPM_LIBRARY=$PWD/build_variants/stamps.so timeout -k 10 300 python -u tools/clock_stamps.py > gpurun_out/$TAG/clock_stamps.json 2>> gpurun_out/$TAG/bench.err
fi
python - "$TAG" <<'PY'
import json, sys, glob
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % sys.argv[1])):
    d = json.load(open(f))
    k = d["kernels"]
    print("%-40s %6.2f G/s %7.4f ms | steady %6.2f G/s rebins %s (%.3f ms/step) | %s" % (
        f.split("/")[-1][6:-5], d["value"] / 1e9, d["ms_per_step"], d["steady"]["value"] / 1e9, d["steady"]["rebins"],
        d["steady"]["rebin_ms_per_step"], {n: x["avg_ms"] for n, x in k.items()}))
    print(" " * 40, "steady kernels", d["steady"].get("kernel_avg_ms"), "stragglers/step", d["steady"].get("stragglers_per_step"))
PY
cat gpurun_out/$TAG/clock_stamps.json 2>/dev/null || true
