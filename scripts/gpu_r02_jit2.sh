#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 600000 1000000 2000000; do
  for v in "1 1" "1 0" "0 1"; do
    set -- $v
    echo "generic n=$n jit=$1 nt=$2 $(GGRS_TICK_GENERIC=1 GGRS_TICK_JIT=$1 GGRS_TICK2_NT=$2 timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab_nt.txt
  done
done
