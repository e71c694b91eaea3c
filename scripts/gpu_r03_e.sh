#!/bin/bash
# round 3, call E: saddr stores in the generated kernel; persistent-form workgroup size A/B; transport-double fan-out tests
O=gpurun_out/${1:-r03e}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zfanout.py tests/test_gpu_knobs.py tests/test_gpu_schema.py -x -q -m gpu > $O/pytest_some.log 2>&1
tail -5 $O/pytest_some.log
run() { echo "== $1: $(env $2 timeout 120 benches/tick_bench $3 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c60-230)" | tee -a $O/plain.txt; }
for n in 1000000 4000000; do
  run "tick3 n=$n" "A=1" $n
  run "jit_tiles n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0" $n
  for t in 256 512 1024; do run "jit_persist tpb=$t n=$n" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_TPB=$t" $n; done
done
run "jit_tiles fullcopy 1M" "GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0 GGRS_ROW_VERSIONS=0" 1000000
run "jit_persist fullcopy 1M" "GGRS_TICK_GENERIC=1 GGRS_ROW_VERSIONS=0" 1000000
for n in 10000 100000 300000; do run "default n=$n" "A=1" $n; done
