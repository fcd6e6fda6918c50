#!/bin/bash
# Timing experiments with parts of g2p and of the momentum pass removed (-DFLOWMPM_ABLATE=<bits>; the results
# are WRONG, only the launch durations of the first steps mean anything).  Build first, on the CPU box:
#   tools/variants.sh base "" abl1 "-DFLOWMPM_ABLATE=1" abl2 "-DFLOWMPM_ABLATE=2" abl4 "-DFLOWMPM_ABLATE=4" \
#       abl5 "-DFLOWMPM_ABLATE=5" abl32 "-DFLOWMPM_ABLATE=32" abl16 "-DFLOWMPM_ABLATE=16"
bash tools/ab.sh "--steady-windows 0 --timing-stride 1 --steps 10 --warmup 3" base abl32 abl4 abl5 abl1 abl2 abl16 base
