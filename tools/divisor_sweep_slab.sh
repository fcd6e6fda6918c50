#!/bin/bash
# Time per step of slabs of the 8-slab decomposition (tools/slab_balance.py --only) for several drift
# divisors of the re-bin rule: tools/divisor_sweep_slab.sh "<slabs>" d1 d2 ...
cd $GRAFT_REPO_ROOT
SLABS=$1; shift
mkdir -p gpurun_out/bal
for r in $SLABS; do
for d in "$@"; do
    FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 300 python3 tools/slab_balance.py --world 8 --only $r --warmup 20 --steps 60 2> gpurun_out/bal/sweep.err | grep "^slab" | sed "s/^/divisor $d: /" || exit 1
done
done
