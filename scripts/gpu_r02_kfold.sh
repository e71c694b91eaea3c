#!/bin/bash
# in-launch two-level checksum fold of the generated kernel: parity at HBM sizes, A/B by size
set -u
OUT=gpurun_out/r02kf; mkdir -p $OUT; export TMPDIR=/tmp
GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py -m gpu -x -q > $OUT/pytest_generic.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_generic.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_generic.txt | tail -n 3
GGRS_TICK_GENERIC=1 GGRS_JIT_FOLD_MIN_WGS=0 GGRS_HOST_FOLD_MAX_WGS=0 GGRS_TICK1_DP=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py -m gpu -x -q > $OUT/pytest_kfold_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_kfold_all.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_kfold_all.txt | tail -n 3
for n in 300000 600000 1000000 2000000 4000000; do
  for f in 1024 1000000000; do
    echo "generic n=$n fold_min=$f $(GGRS_TICK_GENERIC=1 GGRS_JIT_FOLD_MIN_WGS=$f timeout 120 benches/tick_bench $n 8 200 30 0 0 1 2>&1 | tail -n 1 | cut -c1-230)" | tee -a $OUT/ab.txt
  done
done
