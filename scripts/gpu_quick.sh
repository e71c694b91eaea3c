#!/bin/bash
# Short GPU check: parity tests (with durations) + bench lines.  Usage: gpurun -- 'bash scripts/gpu_quick.sh tag'
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
timeout 300 python bench.py --no-cpu-baseline --nt > $OUT/bench_nt.json 2>> $OUT/bench.err; cat $OUT/bench_nt.json
timeout 300 python bench.py --no-cpu-baseline --no-groups > $OUT/bench_nogroups.json 2>> $OUT/bench.err
timeout 300 python bench.py --entities 10000 --no-cpu-baseline > $OUT/bench_10k.json 2>> $OUT/bench.err; cat $OUT/bench_10k.json
timeout 300 python bench.py --entities 100000 --no-cpu-baseline > $OUT/bench_100k.json 2>> $OUT/bench.err; cat $OUT/bench_100k.json
tail -5 $OUT/bench.err
