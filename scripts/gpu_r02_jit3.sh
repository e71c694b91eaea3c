#!/bin/bash
# generated kernel, 4 slots per lane: parity (forced everywhere) + A/B by size
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
GGRS_JIT_V=4 GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest_v4.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_v4.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_v4.txt | tail -n 4
for n in 300000 600000 1000000 2000000; do
  for v in 1 4; do
    echo "generic n=$n v=$v $(GGRS_TICK_GENERIC=1 GGRS_JIT_V=$v timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab_v4.txt
  done
done
echo "generic n=1000000 v=4 nt=0 $(GGRS_TICK_GENERIC=1 GGRS_JIT_V=4 GGRS_TICK2_NT=0 timeout 120 benches/tick_bench 1000000 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab_v4.txt
