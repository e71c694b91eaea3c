#!/bin/bash
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_parity.log; cat $OUT/pytest_parity.log
TB="./benches/tick_bench 1000000 8 200 16 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
run GGRS_TICK2=0
for ilv in 0 1; do for nt in 1 0; do for wgs in 0 1 2 3; do run GGRS_TICK2_ILV=$ilv GGRS_TICK2_NT=$nt GGRS_TICK2_WGS=$wgs; done; done; done
run GGRS_TICK2_ILV=0 GGRS_TICK2_NT=1 GGRS_TICK2_WGS=0 GGRS_TICK_LDS=150000
run GGRS_TICK2_ILV=0 GGRS_TICK2_NT=1 GGRS_TICK2_WGS=0 GGRS_TICK_LDS=76000
