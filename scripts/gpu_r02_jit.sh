#!/bin/bash
# generated request-group kernel (hiprtc): parity suites with it on (default) and off, A/B vs k_tick_gen
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
grep -E "passed|failed|rc=|Error|error" $OUT/pytest.txt | tail -n 6
GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py -m gpu -x -q > $OUT/pytest_generic.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_generic.txt
grep -E "passed|failed|rc=" $OUT/pytest_generic.txt | tail -n 3
for n in 10000 100000 300000 1000000; do
  for jit in 1 0; do
    echo "generic n=$n jit=$jit $(GGRS_TICK_GENERIC=1 GGRS_TICK_JIT=$jit timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab.txt
  done
done
