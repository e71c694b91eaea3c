#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python -m pytest "tests/test_gpu_parity.py::test_particles_synctest_checksums_and_state" -q > $OUT/single_$i.txt 2>&1
  echo "single $i: $(grep -E 'passed|failed' $OUT/single_$i.txt | tail -n 1) $(grep FAILED $OUT/single_$i.txt | tr '\n' ' ')" | tee -a $OUT/single.txt
done
