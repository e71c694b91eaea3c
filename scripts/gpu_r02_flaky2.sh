#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests -m gpu -q > $OUT/rep_$i.txt 2>&1
  echo "rep $i: $(grep -E 'passed|failed' $OUT/rep_$i.txt | tail -n 1) $(grep FAILED $OUT/rep_$i.txt | tr '\n' ' ')" | tee -a $OUT/reps.txt
done
