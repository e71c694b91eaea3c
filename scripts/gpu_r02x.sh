#!/bin/bash
OUT=gpurun_out/${1:-r02x}; mkdir -p $OUT; export TMPDIR=/tmp
for e in "GGRS_DEAD_GROUPS=1" "GGRS_DEAD_GROUPS=0"; do
  env $e timeout 600 python bench.py --fanout --entities 100000 --branches 256 --steps 20 --warmup 3 --no-cpu-baseline 2> $OUT/err_$e.txt | grep '^{' > $OUT/config5_$e.json
  python -c "
import json,sys; j=json.load(open('$OUT/config5_$e.json')); print('$e', 'value %.2f G ef/s' % (j['value']/1e9), 'ms/step %.3f' % j['ms_per_step'], 'branches/s %.0f' % (256/(j['ms_per_step']*1e-3)), j['roofline']['launches_per_step'])"
done
