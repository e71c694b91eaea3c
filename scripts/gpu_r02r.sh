#!/bin/bash
OUT=gpurun_out/${1:-r02r}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zfanout.py -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -30 > $OUT/pytest.log; cat $OUT/pytest.log
