#!/bin/bash
# A/B of the k_tick tile width (GGRS_TICK_VEC=1|4) across world sizes, after a golden parity pass for both.
OUT=gpurun_out/${1:-vecab}; mkdir -p $OUT; export TMPDIR=/tmp
for v in 1 4; do
  GGRS_TICK_VEC=$v timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
  for n in 10000 100000 300000 1000000 4000000; do
    GGRS_TICK_VEC=$v timeout 300 python bench.py --no-cpu-baseline --entities $n 2>>$OUT/bench.err | tee -a $OUT/bench_vec$v.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('vec=$v n=$n', 'G ef/s=%.2f ms/step=%.4f kernel_us=%.1f frac=%.3f' % (d['value']/1e9, d['ms_per_step'], r['avg_launch_us'], r['frac']))"
  done
done
