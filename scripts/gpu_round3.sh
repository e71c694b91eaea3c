#!/bin/bash
# One gpurun call (round 3): GPU parity tests, smoke, the bench lines (headline in the driver's exact form and in the long form,
# BASELINE configs 2 / 4 / 5, the reference's full POD schema, the full-copy variant), rocprofv3 kernel-trace stats of the bench
# command and its FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes), tick_bench size sweep.
# Usage: gpurun -- 'bash scripts/gpu_round3.sh [tag]'
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; echo "bench (driver form) rc=$?"; cut -c1-600 $OUT/bench_driver_form.json
timeout 600 python bench.py > $OUT/bench.json 2>> $OUT/bench.err; echo "bench rc=$?"
GGRS_ROW_VERSIONS=0 timeout 600 python bench.py --no-cpu-baseline --arena both > $OUT/bench_fullcopy.json 2>> $OUT/bench.err
timeout 600 python bench.py --schema full --no-cpu-baseline --parity-ticks 12 > $OUT/bench_schema_full.json 2>> $OUT/bench.err
timeout 600 python bench.py --schema full > $OUT/bench_schema_full_parity.json 2>> $OUT/bench.err
for c in 2 4; do timeout 600 python bench.py --config $c --no-cpu-baseline > $OUT/bench_config$c.json 2>> $OUT/bench.err; done
timeout 600 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu.json
for n in 100000 300000 600000 4000000; do timeout 300 python bench.py --entities $n --no-cpu-baseline > $OUT/bench_$n.json 2>> $OUT/bench.err; done
timeout 300 python bench.py --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-groups --no-cpu-baseline > $OUT/bench_nogroups.json 2>> $OUT/bench.err
timeout 300 python bench.py --fanout --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_fanout_ws1.json
./scripts/ubench_alu > $OUT/ubench_alu.txt 2>&1
for i in 1 2 3; do ./benches/tick_bench 1000000 8 200 16 0 0 1; done > $OUT/tick_bench_3_processes.txt 2>&1
for n in 10000 100000 300000 1000000 4000000; do ./benches/tick_bench $n 8 200 16 0 0 1; done > $OUT/tick_bench_sizes.txt 2>&1
for n in 1000000 4000000; do GGRS_TICK_JIT=0 ./benches/tick_bench $n 8 200 16 0 0 1; GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=1 ./benches/tick_bench $n 8 200 16 0 0 1; done > $OUT/tick_bench_other_kernels.txt 2>&1
BENCH="python bench.py --steps 100 --warmup 16 --no-cpu-baseline --preheat-ms 0"
# the kernel trace runs the DEFAULT command (pre-heat included): its last 200 tick-shaped launches are the timed region bench.py's own
# HIP events sample -- the two figures must agree (kernel_trace_steady.py: last_200_tick_shaped_launches)
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -f csv -d $OUT/prof_sq -o sq -- $BENCH > $OUT/prof_sq.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
find $OUT -name '*.db' -size +20M -delete
ls $OUT
