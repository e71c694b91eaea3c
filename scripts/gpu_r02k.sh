#!/bin/bash
OUT=gpurun_out/${1:-r02k}; mkdir -p $OUT
./scripts/ubench3 "fan_spec nt" > $OUT/spec_const.txt 2>&1
UB_RANDOM=1 ./scripts/ubench3 "fan_spec nt" > $OUT/spec_random.txt 2>&1
UB_RANDOM=1 ./scripts/ubench3 "fronts=9 grid=256 " > $OUT/fill_random.txt 2>&1
