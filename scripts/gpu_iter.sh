#!/bin/bash
# Fast iteration: all gpu tests (fail-fast) + the bench lines at 1M / 100k / 10k, async and sync.
# Usage: gpurun -- 'bash scripts/gpu_iter.sh tag [extra bench args]'
TAG=${1:-iter}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
for cfg in "" "--sync" "--entities 100000" "--entities 100000 --sync" "--entities 10000" "--entities 4000000"; do
  timeout 300 python bench.py --no-cpu-baseline $cfg "$@" 2>>$OUT/bench.err | tee -a $OUT/bench_lines.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('[$cfg]', 'G ef/s=%.2f ms/step=%.4f kernel_us=%.1f frac=%.3f fin_us=%.1f' % (d['value']/1e9, d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('other_kernels',{}).get('k_tick_finalize',{}).get('avg_launch_us',0)))"
done
tail -3 $OUT/bench.err
