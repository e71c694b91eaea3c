#!/bin/bash
# round 6: a wider one-off fuzz of the round's new paths
out=gpurun_out/r06bb; mkdir -p $out
timeout 1500 python scripts/fuzz_more.py branches 8000 8144 > $out/fuzz_branches.txt 2>&1; tail -3 $out/fuzz_branches.txt | cut -c1-700
timeout 1200 python scripts/fuzz_more.py devspawn 9100 9160 > $out/fuzz_devspawn.txt 2>&1; tail -3 $out/fuzz_devspawn.txt | cut -c1-700
