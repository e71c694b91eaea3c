#!/bin/bash
# round 3, call G: persistent form with an oversubscribed grid
O=gpurun_out/${1:-r03g}; mkdir -p $O
run() { echo "== $1: $(env $2 timeout 120 benches/tick_bench $3 8 ${4:-100} 16 0 ${5:-0} 1 2>&1 | tail -n 1 | cut -c60-230)" | tee -a $O/plain.txt; }
for n in 1000000 4000000; do
  run "jit_tiles n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0" $n
  for o in 1 2 4 8; do
    for t in 512 1024; do run "jit_persist oversub=$o tpb=$t n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_OVERSUB=$o GGRS_JIT_PERSIST_TPB=$t" $n; done
  done
done
