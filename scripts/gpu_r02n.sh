#!/bin/bash
OUT=gpurun_out/${1:-r02n}; mkdir -p $OUT
for n in 2000000 4000000 8000000; do for c in 0 1; do
  echo "== n=$n CONTIG=$c" | tee -a $OUT/ab.txt; GGRS_ARENA_CONTIG=$c timeout 120 ./benches/tick_bench $n 8 60 8 0 0 2 2>&1 | tee -a $OUT/ab.txt
done; done
for n in 10000 100000 600000 1000000; do for c in 0 1; do
  echo "== n=$n CONTIG=$c" | tee -a $OUT/ab.txt; GGRS_ARENA_CONTIG=$c timeout 120 ./benches/tick_bench $n 8 200 16 0 0 2 2>&1 | tee -a $OUT/ab.txt
done; done
