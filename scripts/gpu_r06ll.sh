#!/bin/bash
# round 6, last tree: the whole GPU suite + smoke + the driver's bench command (short: no counter passes)
OUT=gpurun_out/r06ll; mkdir -p $OUT
timeout 560 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 60 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
