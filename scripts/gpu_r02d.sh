#!/bin/bash
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
TB="./benches/tick_bench 1000000 8 120 12 0 0 6"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
run GGRS_DEBUG_ARENA=1
run GGRS_DEBUG_ARENA=1 GGRS_ARENA_CONTIG=1
for pad in 4096 65536 262144 1052672 2097152; do run GGRS_ARENA_CONTIG=1 GGRS_BLOCK_PAD=$pad; done
run GGRS_ARENA_CONTIG=1 GGRS_ARENA_ALIGN=2097152
run GGRS_ARENA_CONTIG=1 GGRS_ARENA_ALIGN=1073741824
