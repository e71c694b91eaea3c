#!/bin/bash
# the fallback paths still pass the parity suites: no generated kernel, no depth-parallel roles, forced generic without JIT
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$name.txt 2>&1; echo "$name rc=$? $(grep -E 'passed|failed' $OUT/pytest_$name.txt | tail -n 1)" | tee -a $OUT/summary.txt; }
run default A=1
run nojit GGRS_TICK_JIT=0
run nojit_generic GGRS_TICK_JIT=0 GGRS_TICK_GENERIC=1
run nodp GGRS_TICK1_DP=0 GGRS_GEN_DP=0
run jit_generic_dp2 GGRS_TICK_GENERIC=1 GGRS_TICK1_DP=2
