#!/bin/bash
# What decides the two k_tick modes (~125 us vs ~134 us at 1M)?  arena base alignment / skew vs block stride.
TAG=${1:-pad}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export GGRS_DEBUG_ARENA=1
run() {
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $EXTRA 2>$OUT/one.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('$*', 'G=%.2f kernel_us=%.1f' % (d['value']/1e9, r['avg_launch_us']))" | tee -a $OUT/sweep.txt
  grep "ggrs arena" $OUT/one.err | head -1 | tee -a $OUT/sweep.txt
}
run A=1
run A=2
run GGRS_BLOCK_PAD=4096
run GGRS_BLOCK_PAD=1052672
A2=2097152
for pad in 0 4096 32768 65536 69632; do
  for skew in 0 4096 65536; do run GGRS_ARENA_ALIGN=$A2 GGRS_ARENA_SKEW=$skew GGRS_BLOCK_PAD=$pad; done
done
run GGRS_ARENA_ALIGN=1073741824 GGRS_BLOCK_PAD=0
run GGRS_ARENA_ALIGN=1073741824 GGRS_BLOCK_PAD=4096
