#!/bin/bash
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
GGRS_TICK3=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.log; cat $OUT/pytest.log
TB="./benches/tick_bench 1000000 8 150 12 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" timeout 120 $TB 2>&1 | tee -a $OUT/ab.txt; }
for c in 0 1; do
  for t3 in 2 1; do for wgs in 0 2; do run GGRS_ARENA_CONTIG=$c GGRS_TICK3=$t3 GGRS_TICK2_WGS=$wgs; done; done
done
for e in "GGRS_TICK3=2 GGRS_TICK2_WGS=0" "GGRS_TICK3=1 GGRS_TICK2_WGS=0"; do echo "== 4M $e" | tee -a $OUT/ab.txt; env $e timeout 120 ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
