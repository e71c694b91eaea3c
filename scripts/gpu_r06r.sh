#!/bin/bash
# round 6, call r: user-written systems that cannot defer a despawn no longer count as live-only state (lazy live block, batching, branch steps open to them): whole suite
out=gpurun_out/r06r; mkdir -p $out
timeout 1800 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -60 > $out/pytest_gpu.log; echo "pytest rc=$?"; tail -30 $out/pytest_gpu.log | cut -c1-600
