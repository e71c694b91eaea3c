#!/bin/bash
# host-side fold of small groups: parity suites, then A/B by size
set -u
OUT=gpurun_out/r02hf; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_all.txt | tail -n 3
GGRS_HOST_FOLD_MAX_WGS=100000 GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_zfanout.py -m gpu -x -q > $OUT/pytest_hf_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_hf_all.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_hf_all.txt | tail -n 3
for n in 10000 30000 100000 200000 300000 400000; do
  for hf in 0 100000; do
    echo "n=$n hostfold=$hf $(GGRS_HOST_FOLD_MAX_WGS=$hf timeout 120 benches/tick_bench $n 8 400 50 0 0 1 2>&1 | tail -n 1 | cut -c1-230)" | tee -a $OUT/ab.txt
  done
done
for hf in 0 100000; do
  echo "n=10000 sync hostfold=$hf $(GGRS_HOST_FOLD_MAX_WGS=$hf timeout 120 benches/tick_bench 10000 8 400 50 0 1 1 2>&1 | tail -n 1 | cut -c1-230)" | tee -a $OUT/ab.txt
done
