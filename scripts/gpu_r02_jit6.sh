#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 300000 350000 400000 450000 500000; do
  echo "n=$n jit $(GGRS_JIT_PARTICLES_MAX_SLOTS=9999999 timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/cross.txt
  echo "n=$n tick $(GGRS_JIT_PARTICLES_MAX_SLOTS=0 timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/cross.txt
  echo "n=$n tick3 $(GGRS_JIT_PARTICLES_MAX_SLOTS=0 GGRS_TICK2_MIN_SLOTS=100000 timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/cross.txt
done
