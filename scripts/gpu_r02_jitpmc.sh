#!/bin/bash
# counter passes (TCC / SQ) of the generated kernel on the headline world, next to k_tick3 on the same box
set -u
OUT=gpurun_out/r02jp; mkdir -p $OUT/jit $OUT/tick3; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1
GGRS_TICK_GENERIC=1 PMC_MAX_PASSES=9 timeout 900 python scripts/pmc_passes.py $OUT/jit $OUT/counters.txt -- ./benches/tick_bench 1000000 8 40 8 0 1 1 > $OUT/jit_passes.log 2>&1
PMC_MAX_PASSES=9 timeout 900 python scripts/pmc_passes.py $OUT/tick3 $OUT/counters.txt -- ./benches/tick_bench 1000000 8 40 8 32 1 1 > $OUT/tick3_passes.log 2>&1
rm -rf $OUT/jit/pmc_* $OUT/tick3/pmc_*
ls $OUT/jit $OUT/tick3
