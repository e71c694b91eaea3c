#!/bin/bash
OUT=gpurun_out/${1:-r02v}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -4
GGRS_TICK_GENERIC=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synctest_checksums or headline or p2p" 2>&1 | tail -3
for n in 100000 1000000 4000000; do GGRS_TICK_GENERIC=1 ./benches/tick_bench $n 8 100 12 0 0 1; done 2>&1 | tee $OUT/tb_gen.txt
./benches/tick_bench 1000000 8 200 16 0 0 1
B="./benches/tick_bench 1000000 8 60 8 0 0 1"
GGRS_TICK_GENERIC=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/gen_stats -o s -- $B > $OUT/gen_stats.log 2>&1
GGRS_TICK_GENERIC=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o f -- $B > /dev/null 2>&1
GGRS_TICK_GENERIC=1 timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o w -- $B > /dev/null 2>&1
head -3 $OUT/gen_stats/s_kernel_stats.csv
