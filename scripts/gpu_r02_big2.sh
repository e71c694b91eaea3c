#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 300000 600000 1000000 2000000; do
  for c in 1 0; do
    for nt in 1 0; do
      echo "n=$n jit contig=$c nt=$nt $(GGRS_TICK_GENERIC=1 GGRS_ARENA_CONTIG=$c GGRS_TICK2_NT=$nt timeout 120 benches/tick_bench $n 8 150 16 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/big2.txt
    done
  done
done
