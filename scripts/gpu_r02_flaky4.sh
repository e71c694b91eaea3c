#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
ls -la ~/.cache 2>/dev/null | head; 
rep() { name=$1; shift; "$@" > $OUT/f4_$name.txt 2>&1; echo "$name: $(grep -E 'passed|failed' $OUT/f4_$name.txt | tail -n 1) $(grep FAILED $OUT/f4_$name.txt | tr '\n' ' ')" | tee -a $OUT/f4.txt; }
rep cold1 timeout 900 python -m pytest tests -m gpu -q
ls -la ~/.cache 2>/dev/null | head; du -sh ~/.cache/* 2>/dev/null | head
rep warm2 timeout 900 python -m pytest tests -m gpu -q
rm -rf ~/.cache/comgr ~/.cache/comgr_cache 2>/dev/null
rep nocache3 env AMD_COMGR_CACHE=0 timeout 900 python -m pytest tests -m gpu -q
rep nocache4 env AMD_COMGR_CACHE=0 timeout 900 python -m pytest tests -m gpu -q
