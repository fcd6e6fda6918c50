#!/bin/bash
# usage: pmc.sh <tag> <counters...>   (runs from repo root on the GPU box)
tag=$1; shift
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 3 > gpurun_out/pmc_$tag.out 2>&1
