#!/bin/bash
# Round 2, GPU call 2: k_tick2 (persistent grid, interleaved store/hash, in-kernel fold) -- parity first, then A/B sweep.
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_zfanout.py -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -15 > $OUT/pytest_parity.log; cat $OUT/pytest_parity.log
TB="./benches/tick_bench 1000000 8 200 16 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
run GGRS_TICK2=0
for nt in 1 0; do for wgs in 0 1 2 3; do run GGRS_TICK2_NT=$nt GGRS_TICK2_WGS=$wgs; done; done
echo "== 4M" | tee -a $OUT/ab.txt
GGRS_TICK2=0 ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt
for wgs in 1 2 3; do GGRS_TICK2_WGS=$wgs ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
