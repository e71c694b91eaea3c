#!/bin/bash
# round 6, call t: device spawns -- value tags forced on for some seeds; parents that spawn in consecutive frames (records per step parity)
out=gpurun_out/r06t; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_device_spawn.py -q -m gpu 2>&1 | tail -25 | tee $out/pytest_devspawn.log | cut -c1-900
