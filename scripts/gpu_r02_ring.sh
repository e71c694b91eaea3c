#!/bin/bash
set -u
OUT=gpurun_out/r02hf; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_ring.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ring.txt
grep -E "passed|failed|rc=|Error|s call" $OUT/pytest_ring.txt | tail -n 8
echo "n=10000 3000 steps $(timeout 120 benches/tick_bench 10000 8 3000 50 0 0 1 2>&1 | tail -n 1 | cut -c1-230)"
