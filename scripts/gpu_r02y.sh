#!/bin/bash
OUT=gpurun_out/${1:-r02y}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -6
for i in 1 2; do ./benches/tick_bench 1000000 8 200 16 0 0 1; done
