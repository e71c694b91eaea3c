#!/bin/bash
# depth-parallel split of the generated kernel by world size (the thresholds were measured on k_tick1) + the new big generic test
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
for n in 10000 30000 50000 70000 100000 150000 200000 300000; do
  for dp in 0 1 2 3 5; do
    echo "n=$n dp=$dp $(GGRS_TICK1_DP=$dp GGRS_TICK1_DP_MAX_SLOTS=409600 timeout 120 benches/tick_bench $n 8 400 50 0 0 1 2>&1 | tail -n 1 | cut -c1-230)" | tee -a $OUT/jit_dp.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_gen_groups.py -m gpu -q 2>&1 | tail -n 3
