#!/bin/bash
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_all.txt | tail -n 4
GGRS_JIT_V=4 GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py -m gpu -x -q > $OUT/pytest_v4.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_v4.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_v4.txt | tail -n 3
for n in 10000 100000 300000; do
  echo "default n=$n $(timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/sizes_wgparts.txt
done
for n in 1000000 4000000; do
  echo "generic n=$n $(GGRS_TICK_GENERIC=1 timeout 120 benches/tick_bench $n 8 200 30 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/sizes_wgparts.txt
done
for i in 1 2 3; do
  timeout 600 python bench.py --fanout --entities 100000 --branches 256 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('config5 ms/step %.3f' % j['ms_per_step'])" | tee -a $OUT/sizes_wgparts.txt
done
