#!/bin/bash
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_zfanout.py -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -6 > $OUT/pytest.log; cat $OUT/pytest.log
TB="./benches/tick_bench 1000000 8 150 12 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
for c in 0 1; do
  for wgs in 0 1 2; do run GGRS_ARENA_CONTIG=$c GGRS_TICK3=1 GGRS_TICK2_WGS=$wgs; done
  run GGRS_ARENA_CONTIG=$c GGRS_TICK3=1 GGRS_TICK2_WGS=0 GGRS_TICK2_NT=0
  run GGRS_ARENA_CONTIG=$c GGRS_TICK3=0 GGRS_TICK2_WGS=2
  run GGRS_ARENA_CONTIG=$c GGRS_TICK3=0 GGRS_TICK2_WGS=3
done
for e in "GGRS_TICK3=1 GGRS_TICK2_WGS=0" "GGRS_TICK3=1 GGRS_TICK2_WGS=2" "GGRS_TICK3=0 GGRS_TICK2_WGS=2"; do echo "== 4M $e" | tee -a $OUT/ab.txt; env $e ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
