#!/bin/bash
# round 6, call n: branch steps + adoption fuzzed against the oracle (half of the seeds with value tags and the lazy live block forced)
out=gpurun_out/r06n; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_zfuzz_branches.py -q -m gpu 2>&1 | tail -40 > $out/pytest.log; echo "pytest rc=$?"; tail -30 $out/pytest.log | cut -c1-1500
