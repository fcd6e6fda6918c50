#!/bin/bash
# Kernel statistics of one slab of an N-slab decomposition run alone on the GPU (tools/slab_balance.py
# --only): tools/slab_profile.sh <tag> <slab> [more slab_balance.py arguments]
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; SLAB=$2; shift 2
mkdir -p gpurun_out/bal
export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bal/prof_$TAG -- python3 tools/slab_balance.py --world 8 --only $SLAB --warmup 20 --steps 40 "$@" > gpurun_out/bal/$TAG.txt 2> gpurun_out/bal/$TAG.err
grep "^slab" gpurun_out/bal/$TAG.txt
cp $(find gpurun_out/bal/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/bal/${TAG}_kernel_stats.csv
rm -rf gpurun_out/bal/prof_$TAG
python3 - gpurun_out/bal/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 60.0
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
print("GPU time per step %.3f ms" % (tot / steps))
for r in rows[:14]:
    print("  %-46s calls %4s  %7.3f ms per step" % (r['Name'][:46], r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps))
PY
