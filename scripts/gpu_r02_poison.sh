#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
rep() { name=$1; shift; "$@" > $OUT/p_$name.txt 2>&1; echo "$name: $(grep -E 'passed|failed' $OUT/p_$name.txt | tail -n 1) $(grep FAILED $OUT/p_$name.txt | head -n 12 | tr '\n' ' ')" | tee -a $OUT/poison.txt; }
rep poison_jit env GGRS_DEBUG_POISON=1 timeout 900 python -m pytest tests -m gpu -q
rep poison_nojit env GGRS_DEBUG_POISON=1 GGRS_TICK_JIT=0 timeout 900 python -m pytest tests -m gpu -q
