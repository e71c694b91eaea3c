#!/bin/bash
# k_tick_gen: slots per workgroup by world size
set -u
OUT=gpurun_out/r02gd; mkdir -p $OUT
for n in 50000 100000 200000 300000 400000 600000 1000000; do
  for sub in 256 512 1024; do
    echo "gen n=$n sub=$sub $(GGRS_TICK_GENERIC=1 GGRS_GEN_DP=0 GGRS_GEN_SUB=$sub timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/sub.txt
  done
done
