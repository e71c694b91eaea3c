#!/bin/bash
# Minimal GPU verification: all gpu tests, smoke, the default bench line (+ --sync and 100k).  Usage: gpurun -- 'bash scripts/gpu_verify.sh tag'
TAG=${1:-verify}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
timeout 600 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -20 $OUT/pytest_gpu.log
timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 200 python bench.py --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err; cat $OUT/bench_sync.json
timeout 200 python bench.py --entities 100000 --no-cpu-baseline > $OUT/bench_100k.json 2>> $OUT/bench.err; cat $OUT/bench_100k.json
tail -5 $OUT/bench.err
