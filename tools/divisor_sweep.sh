#!/bin/bash
# Time per step of one slab (tools/slab_balance.py --only) and of the single-GPU bench for several
# drift divisors of the re-bin rule: tools/divisor_sweep.sh <slab> d1 d2 ...
cd $GRAFT_REPO_ROOT
SLAB=$1; shift
mkdir -p gpurun_out/bal
for d in "$@"; do
    FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 300 python3 tools/slab_balance.py --world 8 --only $SLAB --warmup 20 --steps 60 2> gpurun_out/bal/sweep.err | grep "^slab" | sed "s/^/divisor $d: /" || exit 1
    FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/bal/sweep_bench_$d.json 2>> gpurun_out/bal/sweep.err || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/bal/sweep_bench_$d.json')); print('divisor $d: bench %.3f ms/step %.3e/s' % (d['ms_per_step'], d['value']), d['rebin'])"
done
