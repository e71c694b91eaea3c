#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
./scripts/ubench3 "G=8 " > $OUT/ubench3_spec.txt 2>&1
UB_CONTIG=1 ./scripts/ubench3 "G=8 " > $OUT/ubench3_spec_contig.txt 2>&1
