#!/bin/bash
# round 6, call s: a broadcast adoption the owner cannot serve is refused on EVERY rank (status agreement before the collective); the fan-out file
out=gpurun_out/r06s; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_zfanout.py -q -m gpu 2>&1 | tail -30 | tee $out/pytest_fanout.log | cut -c1-1200
