#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT
for e in "A=1" "GGRS_TICK1_DP=0" "GGRS_TICK_JIT=0" "GGRS_TICK_JIT=0 GGRS_TICK1_DP=0" "GGRS_DEAD_GROUPS=0"; do
  echo "$e: $(env $e timeout 600 python scripts/flaky_probe.py 1 2 12 300 2>&1 | tail -n 1)" | tee -a $OUT/flaky.txt
done
echo "n=63: $(timeout 600 python scripts/flaky_probe.py 63 1 10 200 2>&1 | tail -n 1)" | tee -a $OUT/flaky.txt
echo "n=1000: $(timeout 600 python scripts/flaky_probe.py 1000 2 20 100 2>&1 | tail -n 1)" | tee -a $OUT/flaky.txt
