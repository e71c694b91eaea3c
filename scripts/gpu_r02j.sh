#!/bin/bash
OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT
TB="./benches/tick_bench 1000000 8 150 12 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
for c in 0 1; do
run GGRS_ARENA_CONTIG=$c GGRS_TICK3=1 GGRS_TICK2_WGS=0
run GGRS_ARENA_CONTIG=$c GGRS_TICK3=1 GGRS_TICK2_WGS=0 LD_PRELOAD=scripts/dbg/libggrs_hip.so
run GGRS_ARENA_CONTIG=$c GGRS_TICK3=0 GGRS_TICK2_WGS=3
run GGRS_ARENA_CONTIG=$c GGRS_TICK3=0 GGRS_TICK2_WGS=3 LD_PRELOAD=scripts/dbg/libggrs_hip.so
done
