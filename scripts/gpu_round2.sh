#!/bin/bash
# One gpurun call (round 2): GPU parity tests, smoke, bench lines, rocprofv3 kernel-trace stats of the bench command, and
# the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).
# Usage: gpurun -- 'bash scripts/gpu_round2.sh [tag]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
for n in 10000 100000 300000 600000 4000000; do timeout 300 python bench.py --entities $n --no-cpu-baseline > $OUT/bench_$n.json 2>> $OUT/bench.err; done
timeout 300 python bench.py --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-groups --no-cpu-baseline > $OUT/bench_nogroups.json 2>> $OUT/bench.err
timeout 300 python bench.py --fanout --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_fanout_ws1.json
for i in 1 2 3; do ./benches/tick_bench 1000000 8 200 16 32 0 1; done > $OUT/tick_bench_3_processes.txt 2>&1     # flags = 32: GGRS_WORLD_CONTIG_ARENA
./benches/tick_bench 1000000 8 200 16 0 0 3 > $OUT/tick_bench_paged_arena.txt 2>&1                                    # library default: paged
BENCH="python bench.py --steps 100 --warmup 16 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- $BENCH > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
# BASELINE config 5 on one GPU: 256 predicted-input branches x 100 k entities x 8 frames per step
timeout 600 python bench.py --fanout --entities 100000 --branches 256 --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu.json
# the generated request-group kernel on the headline world (GGRS_TICK_GENERIC=1: as for a world k_tick3 does not cover):
# sizes, then kernel-trace stats and the FETCH / WRITE passes of the 1 M run
for n in 10000 100000 300000 1000000 4000000; do GGRS_TICK_GENERIC=1 ./benches/tick_bench $n 8 200 16 0 0 1; done > $OUT/jit_generic_sizes.txt 2>&1
JB="./benches/tick_bench 1000000 8 100 16 0 0 1"
GGRS_TICK_GENERIC=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_jit_stats -o stats -- $JB > $OUT/prof_jit_stats.log 2>&1
GGRS_TICK_GENERIC=1 timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_jit_fetch -o fetch -- $JB > $OUT/prof_jit_fetch.log 2>&1
GGRS_TICK_GENERIC=1 timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_jit_write -o write -- $JB > $OUT/prof_jit_write.log 2>&1
find $OUT -name '*.db' -size +20M -delete
ls $OUT
