#!/bin/bash
set -u
OUT=gpurun_out/r02cs; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_custom_system.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -n 25 $OUT/pytest.txt
