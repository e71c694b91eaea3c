#!/bin/bash
# generated kernel as the default small-world path: full suite, sizes sweep, config 5
set -u
OUT=gpurun_out/r02jit; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.txt
grep -E "passed|failed|rc=|Error" $OUT/pytest_all.txt | tail -n 4
for n in 10000 30000 100000 300000 480000 600000 1000000; do
  echo "default n=$n $(timeout 120 benches/tick_bench $n 8 300 40 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/sizes_default.txt
done
for e in "GGRS_TICK_JIT=1" "GGRS_TICK_JIT=0"; do
  env $e timeout 600 python bench.py --fanout --entities 100000 --branches 256 --steps 20 --warmup 3 --no-cpu-baseline 2> $OUT/err_$e.txt | grep '^{' > $OUT/config5_$e.json
  python -c "
import json,sys; j=json.load(open('$OUT/config5_$e.json')); print('$e', 'value %.2f G ef/s' % (j['value']/1e9), 'ms/step %.3f' % j['ms_per_step'], 'branches/s %.0f' % (256/(j['ms_per_step']*1e-3)), j['roofline'].get('launches_per_step'), j.get('parity'))" | tee -a $OUT/config5.txt
done
