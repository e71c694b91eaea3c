#!/bin/bash
set -u
OUT=gpurun_out/r02fc; mkdir -p $OUT; export TMPDIR=/tmp
echo "boot $(cat /proc/sys/kernel/random/boot_id | cut -c1-8)"
timeout 900 python scripts/flaky_diag.py > $OUT/diag.txt 2>&1
grep "^\[exact\|^\[first\]\|^\[second\]\|^    \|prefix rc" $OUT/diag.txt | grep -v "equal True" | cut -c1-600
