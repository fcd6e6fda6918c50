#!/bin/bash
# The round-3 collection (on the GPU box, from the repository root): tools/r03_round.sh <tag>
#   bench lines of every single-GPU configuration, rocprofv3 kernel statistics of the default line, the four
#   separate --pmc passes, the slab step on eight slabs in one process, the C-ABI driver's output.
set -e
TAG=${1:-r03a}
out=gpurun_out/$TAG
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $out/bench_16M_256cube.json 2> $out/bench.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats_16M_256cube.csv
# the driver's own command under the profiler: its steady roofline must reproduce from these statistics
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_cmd_under_rocprof.json 2>> $out/rocprof.err
cp $(find $out/stats_driver -name "*kernel_stats.csv" | head -1) $out/kernel_stats_driver_cmd.csv
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2>> $out/bench.err
bash tools/pmc.sh ${TAG}_fetch TCC_EA0_RDREQ_32B_sum FETCH_SIZE
bash tools/pmc.sh ${TAG}_write WRITE_SIZE TCC_EA0_WRREQ_64B_sum
bash tools/pmc.sh ${TAG}_sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
bash tools/pmc.sh ${TAG}_atomic TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum
for t in fetch write sq atomic; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$t > $out/pmc_$t.txt; done
timeout -k 10 300 python bench.py --no-cpu-baseline --rounds 5 > $out/bench_16M_256cube_rounds5.json 2>> $out/bench.err
bash tools/other_configs.sh $TAG
timeout -k 10 300 python -u tools/slab_group_bench.py --world 8 --steps 40 --windows 2 --single --profile-slab 3 > $out/slab_group_8slabs.json 2>> $out/bench.err
timeout -k 10 200 python bench.py --one-slab --no-cpu-baseline > $out/bench_one_slab.json 2>> $out/bench.err
echo round done
