# Sweep of the re-binning policy: SWEEP_ARGS selects the workload, then "interval divisor" pairs.
mkdir -p gpurun_out/sweep
for cfg in $SWEEP_CFGS; do i=${cfg%:*}; d=${cfg#*:}
FLOWMPM_DRIFT_DIVISOR=$d timeout -k 10 150 python bench.py --rebin-interval $i --no-cpu-baseline $SWEEP_ARGS > gpurun_out/sweep/run.json 2> gpurun_out/sweep/run.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/sweep/run.json')); print('$SWEEP_ARGS interval $i divisor $d', round(d['value']/1e9,3), d['rebin'], d['kernels']['g2p']['avg_ms'], d['kernels']['p2g']['avg_ms'])"
done
