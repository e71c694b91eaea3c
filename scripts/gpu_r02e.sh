#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GGRS_ARENA_CONTIG=1
TB="./benches/tick_bench 1000000 8 150 12 0 0 2"
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" $TB 2>&1 | tee -a $OUT/ab.txt; }
run GGRS_TICK2=0
run GGRS_TICK2=0 GGRS_TICK_REST=0
for ilv in 0 1; do for nt in 1 0; do for wgs in 0 1 2 3; do run GGRS_TICK2_ILV=$ilv GGRS_TICK2_NT=$nt GGRS_TICK2_WGS=$wgs; done; done; done
for lds in 150000 76000 52000; do run GGRS_TICK2_ILV=0 GGRS_TICK2_NT=1 GGRS_TICK2_WGS=0 GGRS_TICK_LDS=$lds; done
echo "== 4M" | tee -a $OUT/ab.txt
for e in "GGRS_TICK2=0" "GGRS_TICK2_WGS=0" "GGRS_TICK2_WGS=2" "GGRS_TICK2_WGS=3" "GGRS_TICK2_WGS=3 GGRS_TICK2_ILV=1"; do echo "== 4M $e" | tee -a $OUT/ab.txt; env $e ./benches/tick_bench 4000000 8 60 8 0 0 1 2>&1 | tee -a $OUT/ab.txt; done
UB_CONTIG=1 ./scripts/ubench3 "fan " > $OUT/ubench3_fan_contig.txt 2>&1
UB_CONTIG=1 ./scripts/ubench3 "fronts=9" > $OUT/ubench3_fill9_contig.txt 2>&1
./scripts/ubench3 "fronts=9 grid=256 " > $OUT/ubench3_fill9_plainalloc.txt 2>&1
