set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r02c}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > gpurun_out/$TAG/bench_16M_256cube.json 2> gpurun_out/$TAG/bench.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/rocprof.err
bash tools/pmc.sh ${TAG}_fetch TCC_EA0_RDREQ_32B_sum FETCH_SIZE
bash tools/pmc.sh ${TAG}_write WRITE_SIZE TCC_EA0_WRREQ_64B_sum
bash tools/pmc.sh ${TAG}_sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
bash tools/pmc.sh ${TAG}_atomic TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum
for t in fetch write sq atomic; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$t > gpurun_out/$TAG/pmc_$t.txt; done
timeout -k 10 300 python bench.py --no-cpu-baseline --rounds 5 > gpurun_out/$TAG/bench_rounds5.json 2>> gpurun_out/$TAG/bench.err
cp $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats_16M_256cube.csv
echo done
# the reference's default noise: kernel trace of the steady state (400 steps before the trace would be better; the trace covers all)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_noise1 -- python3 bench.py --no-cpu-baseline --noise 1.0 > gpurun_out/$TAG/bench_noise1_under_rocprof.json 2>> gpurun_out/$TAG/rocprof.err
cp $(find gpurun_out/$TAG/stats_noise1 -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats_16M_256cube_noise1.csv
