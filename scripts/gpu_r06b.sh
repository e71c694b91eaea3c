#!/bin/bash
# round 6, call b: why is the spawn world's branch launch slower?  kernel traces of config 5 with / without the spawn system, with / without firing spawns
TAG=${1:-r06b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --config 5 --steps 10 --warmup 2 --preheat-ms 0 --no-cpu-baseline"
for v in nospawn spawn spawn_zero; do
  case $v in nospawn) X=""; E="";; spawn) X="--spawn"; E="";; spawn_zero) X="--spawn"; E="BENCH_BRANCH_INPUT_ZERO=1";; esac
  env $E timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$v -o t -- $B $X > $OUT/prof_$v.log 2>&1
  f=$(find $OUT/prof_$v -name '*kernel_stats.csv' | head -1); echo "== $v"; head -4 $f | cut -c1-200
done
find $OUT -name '*.db' -size +20M -delete
