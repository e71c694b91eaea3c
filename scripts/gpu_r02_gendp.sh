#!/bin/bash
# depth-parallel roles in k_tick_gen: A/B by world size and outputs per role, then the generic parity tests
set -u
OUT=gpurun_out/r02gd; mkdir -p $OUT
for n in 10000 50000 100000 200000 400000 1000000; do
  for dp in 0 1 2 3 5; do
    echo "gen n=$n dp=$dp $(GGRS_TICK_GENERIC=1 GGRS_GEN_DP=$dp GGRS_GEN_DP_MAX_SLOTS=2000000 timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/ab.txt
  done
done
GGRS_TICK_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_despawn_rollback.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
GGRS_TICK_GENERIC=1 GGRS_GEN_DP=1 GGRS_GEN_DP_MAX_SLOTS=2000000 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_box_game.py -m gpu -x -q > $OUT/pytest_dp1.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_dp1.txt
for f in $OUT/pytest.txt $OUT/pytest_dp1.txt; do grep -E "passed|failed|rc=" $f | tail -n 2; done
