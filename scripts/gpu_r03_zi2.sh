#!/bin/bash
OUT=gpurun_out/r03zi; mkdir -p $OUT
{
GGRS_JIT_SPECIALISE_AFTER=4 GGRS_JIT_SPECIALISE_SYNC=1 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=0 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=4 GGRS_JIT_SPECIALISE_SYNC=1 GGRS_EVENT_ON_KERNEL=0 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=4 GGRS_JIT_SPECIALISE_SYNC=1 GGRS_JIT_LANE_FOLD=0 timeout 100 python scripts/p2p_diag.py
} > $OUT/p2p_diag.txt 2>&1
cat $OUT/p2p_diag.txt
