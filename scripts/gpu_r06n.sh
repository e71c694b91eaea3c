#!/bin/bash
# round 6, call n: device spawns fuzzed against the oracle
out=gpurun_out/r06n; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_device_spawn.py -q -m gpu 2>&1 | tail -40 > $out/pytest_devspawn.log; echo "pytest rc=$?"; tail -30 $out/pytest_devspawn.log | cut -c1-1200
