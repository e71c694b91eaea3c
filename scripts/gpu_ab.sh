#!/bin/bash
# A/B of k_tick launch knobs (env): GGRS_TICK_LDS (occupancy throttle), GGRS_TICK_REST (rest rows stored per Save).
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gen_groups.py tests/test_box_game.py tests/test_cpp_host.py -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -4 | tee $OUT/pytest_gen.log
run() {
  label="$1"; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 300 $BENCH_EXTRA 2>>$OUT/err.log | tee -a $OUT/lines.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('[$label]', 'G ef/s=%.2f ms/step=%.4f kernel_us=%.1f frac=%.3f' % (d['value']/1e9, d['ms_per_step'], r['avg_launch_us'], r['frac']))"
}
run "k_tick 1M" A=1
run "k_tick_gen 1M (LDS-staged generic kernel)" GGRS_TICK_GENERIC=1
BENCH_EXTRA="--entities 100000" run "k_tick1 100k" A=1
BENCH_EXTRA="--entities 100000" run "k_tick_gen 100k" GGRS_TICK_GENERIC=1
BENCH_EXTRA="--entities 10000" run "k_tick1 10k" A=1
BENCH_EXTRA="--entities 10000" run "k_tick_gen 10k" GGRS_TICK_GENERIC=1
BENCH_EXTRA="--entities 100000 --no-groups --sync" run "per-request 100k sync" A=1
BENCH_EXTRA="--entities 10000 --no-groups --sync" run "per-request 10k sync" A=1
tail -3 $OUT/err.log
