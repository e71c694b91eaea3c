#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 1000000 2000000 4000000; do
  for nt in 1 0; do
    echo "generic n=$n v=1 nt=$nt $(GGRS_TICK_GENERIC=1 GGRS_JIT_V=1 GGRS_TICK2_NT=$nt timeout 120 benches/tick_bench $n 8 200 30 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab_v1nt.txt
  done
  echo "generic n=$n gen $(GGRS_TICK_GENERIC=1 GGRS_TICK_JIT=0 timeout 120 benches/tick_bench $n 8 200 30 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab_v1nt.txt
done
