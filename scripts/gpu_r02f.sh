#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1
for cfg in "c1_nt1:GGRS_ARENA_CONTIG=1 GGRS_TICK2_NT=1" "c1_nt0:GGRS_ARENA_CONTIG=1 GGRS_TICK2_NT=0" "c0_nt1:GGRS_ARENA_CONTIG=0 GGRS_TICK2_NT=1" "c0_nt0:GGRS_ARENA_CONTIG=0 GGRS_TICK2_NT=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  mkdir -p $OUT/$name
  env $envs GGRS_TICK2_WGS=3 PMC_MAX_PASSES=8 timeout 600 python scripts/pmc_passes.py $OUT/$name $OUT/counters.txt -- ./benches/tick_bench 1000000 8 30 6 0 1 1 > $OUT/$name/log.txt 2>&1
  env $envs GGRS_TICK2_WGS=3 ./benches/tick_bench 1000000 8 100 10 0 0 1 > $OUT/$name/timing.txt 2>&1
  find $OUT/$name -name '*.csv' -delete
done
