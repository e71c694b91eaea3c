#!/bin/bash
# round 3, call D: DPP wave_xor + straight-line row-mask paths + memoised hash tails -- suite, then the three fused kernels on the headline world
O=gpurun_out/${1:-r03d}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
for v in tick3 jit_tiles jit_persist; do
  case $v in
    tick3) ENVV="A=1";;
    jit_tiles) ENVV="GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0";;
    jit_persist) ENVV="GGRS_TICK_GENERIC=1";;
  esac
  for n in 1000000 4000000; do
    echo "== $v n=$n: $(env $ENVV timeout 120 benches/tick_bench $n 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $O/plain.txt
  done
  echo "== $v n=1M fullcopy: $(env $ENVV GGRS_ROW_VERSIONS=0 timeout 120 benches/tick_bench 1000000 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $O/plain.txt
done
for n in 10000 100000 300000; do
  echo "== default n=$n: $(timeout 120 benches/tick_bench $n 8 200 16 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $O/plain.txt
done
