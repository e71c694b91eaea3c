#!/bin/bash
# A/B of k_tick launch knobs (env): GGRS_TICK_LDS (occupancy throttle), GGRS_TICK_REST (rest rows stored per Save).
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  label="$1"; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 300 $BENCH_EXTRA 2>>$OUT/err.log | tee -a $OUT/lines.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('[$label]', 'G ef/s=%.2f ms/step=%.4f kernel_us=%.1f frac=%.3f' % (d['value']/1e9, d['ms_per_step'], r['avg_launch_us'], r['frac']))"
}
run "default" A=1
run "diag 1: no wave reduction" GGRS_TICK_DIAG=1
run "diag 2: no mask rebuild" GGRS_TICK_DIAG=2
run "diag 4: no partial stores" GGRS_TICK_DIAG=4
run "diag 7: all three" GGRS_TICK_DIAG=7
BENCH_EXTRA="--nt" run "nt" A=1
BENCH_EXTRA="--nt" run "nt diag 7" GGRS_TICK_DIAG=7
BENCH_EXTRA="--nt" run "nt diag 1" GGRS_TICK_DIAG=1
run "default #2" A=1
tail -3 $OUT/err.log
