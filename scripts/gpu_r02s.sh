#!/bin/bash
OUT=gpurun_out/${1:-r02s}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_zfanout.py -m gpu -x -q -rs -k "native" 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -8
timeout 600 python bench.py --fanout --no-cpu-baseline 2> $OUT/err.txt | grep '^{' > $OUT/bench_fanout_ws1.json; cat $OUT/bench_fanout_ws1.json | cut -c1-330; tail -3 $OUT/err.txt
