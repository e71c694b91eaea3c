#!/bin/bash
# round 3, call F: side-stream fold -- suite, then per-tile vs persistent vs k_tick3 per step
O=gpurun_out/${1:-r03f}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
run() { echo "== $1: $(env $2 timeout 120 benches/tick_bench $3 8 ${4:-100} 16 0 ${5:-0} 1 2>&1 | tail -n 1 | cut -c60-230)" | tee -a $O/plain.txt; }
for n in 1000000 4000000; do
  run "tick3 n=$n" "A=1" $n
  run "jit_tiles side-stream n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0" $n
  run "jit_tiles same-stream n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0 GGRS_FIN_SIDE_STREAM=0" $n
  run "jit_persist n=$n" "GGRS_TICK_GENERIC=1" $n
done
run "jit_tiles side-stream SYNC 1M" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0" 1000000 100 1
for n in 100000 300000 600000; do run "default n=$n" "A=1" $n 200; run "same-stream n=$n" "GGRS_FIN_SIDE_STREAM=0" $n 200; done
