#!/bin/bash
# A re-bin every step, riding on the step (default) against the standalone two-launch re-bin, every launch
# timed; then the column-stride experiment.  usage: tools/ride_ab.sh <tag>
TAG=${1:-ride}
mkdir -p gpurun_out/$TAG
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %6.2f G/s %7.4f ms | rebins %3s %.3f ms/step | %s" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"], d["rebin"]["performed"],
      d["rebin"]["ms_per_step"], {n: x["avg_ms"] for n, x in d["kernels"].items() if n in ("p2g1", "p2g2", "g2p")}), "timed rebin launch", d["rebin"]["avg_ms_per_timed_step"])
PY
}
B="timeout -k 10 300 python -u bench.py --no-cpu-baseline --steady-windows 0 --timing-stride 1 --steps 40 --warmup 20"
for noise in 0.0 1.0; do
  $B --noise $noise --rebin-interval 1 > gpurun_out/$TAG/ride_every_step_n$noise.json && show gpurun_out/$TAG/ride_every_step_n$noise.json "ride, every step, noise $noise"
  FLOWMPM_RIDE_REBINS=0 $B --noise $noise --rebin-interval 1 > gpurun_out/$TAG/alone_every_step_n$noise.json && show gpurun_out/$TAG/alone_every_step_n$noise.json "alone, every step, noise $noise"
  FLOWMPM_DRIFT_DIVISOR=0 $B --noise $noise --rebin-interval 100000 > gpurun_out/$TAG/no_rebin_n$noise.json && show gpurun_out/$TAG/no_rebin_n$noise.json "no re-bin, noise $noise"
done
for pad in 0 64 1088 17472; do
  FLOWMPM_BENCH_PAD=$pad timeout -k 10 300 python -u bench.py --no-cpu-baseline --steady-windows 0 > gpurun_out/$TAG/pad_$pad.json && show gpurun_out/$TAG/pad_$pad.json "pad $pad"
done
