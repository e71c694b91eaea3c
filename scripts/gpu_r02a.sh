#!/bin/bash
# Round 2, GPU call 1: store-path microbenchmark, placement A/B of the shipped engine, L2/EA counters of k_tick,
# then the GPU parity tests (incl. the new headline-size oracle tests) and the bench line with its parity gate.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1
timeout 300 ./scripts/ubench3 > $OUT/ubench3.txt 2>&1; echo "ubench3 rc=$?"
GGRS_ARENA_PROBE=0 timeout 120 ./benches/tick_bench 1000000 8 200 16 0 0 5 > $OUT/tb_noprobe.txt 2>&1
timeout 120 ./benches/tick_bench 1000000 8 200 16 0 0 1 > $OUT/tb_probe.txt 2>&1
GGRS_ARENA_PROBE=0 timeout 120 ./benches/tick_bench 1000000 8 200 16 4 0 3 > $OUT/tb_nt.txt 2>&1
GGRS_ARENA_PROBE=0 timeout 120 ./benches/tick_bench 4000000 8 60 8 0 0 2 > $OUT/tb_4m.txt 2>&1
GGRS_ARENA_PROBE=0 GGRS_TICK_GENERIC=1 timeout 120 ./benches/tick_bench 1000000 8 100 8 0 0 1 > $OUT/tb_gen.txt 2>&1
cat $OUT/tb_*.txt
GGRS_ARENA_PROBE=0 PMC_MAX_PASSES=9 timeout 900 python scripts/pmc_passes.py $OUT $OUT/counters.txt -- ./benches/tick_bench 1000000 8 40 8 0 1 3 > $OUT/pmc_passes.log 2>&1
tail -60 $OUT/pmc_passes.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
find $OUT -name '*.db' -delete
