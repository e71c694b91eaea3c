#!/bin/bash
# depth-parallel k_tick1 with per-role in-kernel fold: parity + A/B by world size
set -u
OUT=gpurun_out/r02dp; mkdir -p $OUT
for n in 10000 30000 50000 70000 100000 150000 200000; do
  for dp in 0 1 2 3; do
    echo "n=$n dp=$dp $(GGRS_TICK1_DP=$dp GGRS_TICK1_DP_MAX_SLOTS=409600 timeout 120 benches/tick_bench $n 8 400 50 0 0 1 2>&1 | tail -n 1 | cut -c1-230)" | tee -a $OUT/ab3.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_zfanout.py tests/test_despawn_rollback.py tests/test_cpp_host.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
GGRS_TICK1_DP=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q > $OUT/pytest_dp3.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_dp3.txt
for f in $OUT/pytest.txt $OUT/pytest_dp3.txt; do tail -n 3 $f; done
