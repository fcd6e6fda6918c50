#!/bin/bash
# A/B of library variants under build_variants/ on one box: the parity and slab tests with each variant loaded
# ("base" is taken as known good), then bench lines of every variant, interleaved.
#   tools/variant_ab.sh name1 name2 ...
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
    [ "$v" = base ] && continue
    FLOWMPM_LIBRARY=$PWD/build_variants/$v.so timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py tests/test_slab.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/variant_parity_$v.log 2>&1 || { echo "parity with $v failed"; tail -15 gpurun_out/variant_parity_$v.log; exit 1; }
    echo "$v: $(tail -1 gpurun_out/variant_parity_$v.log)"
done
for round in 1 2; do
    tools/ab_steady.sh "--steps 100 --warmup 20" "$@" || exit 1
done
tools/ab.sh "--steps 100 --warmup 20 --noise 1.0" "$@" || exit 1
tools/ab.sh "--particles-log2 20 --grid 128,128,128 --steps 200 --warmup 40" "$@" || exit 1
tools/ab.sh "--particles 81920 --grid 128,64,64 --steps 400 --warmup 100" "$@" || exit 1
