#!/bin/bash
# A/B of k_tick launch knobs (env): GGRS_TICK_LDS (occupancy throttle), GGRS_TICK_REST (rest rows stored per Save).
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
GGRS_TICK_VEC=1 timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -3
run() {
  label="$1"; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 300 $BENCH_EXTRA 2>>$OUT/err.log | tee -a $OUT/lines.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('[$label]', 'G ef/s=%.2f ms/step=%.4f kernel_us=%.1f frac=%.3f' % (d['value']/1e9, d['ms_per_step'], r['avg_launch_us'], r['frac']))"
}
BENCH_EXTRA="--entities 10000" run "10k k_tick1" A=1
BENCH_EXTRA="--entities 100000" run "100k k_tick1" A=1
BENCH_EXTRA="--entities 300000" run "300k k_tick1" A=1
BENCH_EXTRA="--entities 10000" run "10k k_tick1 #2" A=1
BENCH_EXTRA="--entities 100000" run "100k k_tick1 #2" A=1
tail -3 $OUT/err.log
