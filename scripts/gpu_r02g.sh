#!/bin/bash
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -6 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
TB="./benches/tick_bench 1000000 8 150 12 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
for c in 1 0; do
  run GGRS_ARENA_CONTIG=$c GGRS_TICK2=0
  for nt in 1 0; do for wgs in 0 1 2 3; do run GGRS_ARENA_CONTIG=$c GGRS_TICK2_NT=$nt GGRS_TICK2_WGS=$wgs; done; done
done
for e in "GGRS_TICK2=0" "GGRS_TICK2_WGS=0" "GGRS_TICK2_WGS=2" "GGRS_TICK2_WGS=3" "GGRS_ARENA_CONTIG=1 GGRS_TICK2_WGS=2" "GGRS_ARENA_CONTIG=1 GGRS_TICK2_WGS=3"; do echo "== 4M $e" | tee -a $OUT/ab.txt; env $e ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
for n in 10000 100000 300000 600000; do echo "== n=$n" | tee -a $OUT/ab.txt; ./benches/tick_bench $n 8 200 16 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
./scripts/ubench3 "layout=2" > $OUT/ubench3_G.txt 2>&1
UB_CONTIG=1 ./scripts/ubench3 "layout=2" > $OUT/ubench3_G_contig.txt 2>&1
