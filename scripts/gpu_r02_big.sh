#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 2000000 3000000 4000000 8000000; do
  echo "n=$n tick3 $(timeout 120 benches/tick_bench $n 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/big.txt
  echo "n=$n jit $(GGRS_TICK_GENERIC=1 timeout 120 benches/tick_bench $n 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/big.txt
done
echo "n=4000000 jit_contig $(GGRS_TICK_GENERIC=1 GGRS_ARENA_CONTIG=1 timeout 120 benches/tick_bench 4000000 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/big.txt
echo "n=4000000 jit_nt0 $(GGRS_TICK_GENERIC=1 GGRS_TICK2_NT=0 timeout 120 benches/tick_bench 4000000 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-200)" | tee -a $OUT/big.txt
