#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench line, rocprofv3 kernel-trace stats, and the
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).
# Usage: gpurun -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 300 python bench.py --entities 10000 --no-cpu-baseline > $OUT/bench_10k.json 2>> $OUT/bench.err
timeout 300 python bench.py --entities 100000 --no-cpu-baseline > $OUT/bench_100k.json 2>> $OUT/bench.err
timeout 300 python bench.py --unfused --no-cpu-baseline > $OUT/bench_unfused.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-groups --no-cpu-baseline > $OUT/bench_nogroups.json 2>> $OUT/bench.err
timeout 300 python bench.py --entities 4000000 --no-cpu-baseline > $OUT/bench_4m.json 2>> $OUT/bench.err
timeout 300 python bench.py --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-groups --sync --no-cpu-baseline > $OUT/bench_nogroups_sync.json 2>> $OUT/bench.err
timeout 300 python bench.py --fanout --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_fanout_ws1.json
[ -x scripts/ubench2 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench2.hip -o scripts/ubench2 > /dev/null 2>&1
timeout 120 ./scripts/ubench2 > $OUT/ubench2.txt 2>&1
BENCH="python bench.py --steps 50 --warmup 8 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- $BENCH > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
# keep only csv summaries (sqlite dbs can be large)
find $OUT -name '*.db' -size +20M -delete
ls -laR $OUT | head -80
