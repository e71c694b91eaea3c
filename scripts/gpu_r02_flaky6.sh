#!/bin/bash
# first process on a fresh box: the suite up to the flaky case, arena kinds logged
set -u
TAG=${1:-a}
OUT=gpurun_out/r02fc; mkdir -p $OUT; export TMPDIR=/tmp
GGRS_DEBUG_ARENA=1 timeout 900 python -m pytest tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x -s > $OUT/first_$TAG.txt 2>&1
echo "first $TAG: $(grep -E 'passed|failed' $OUT/first_$TAG.txt | tail -n 1) contiguous=$(grep -c 'contiguous allocation' $OUT/first_$TAG.txt) paged=$(grep -c 'paged allocation' $OUT/first_$TAG.txt) $(hostname) $(cat /proc/sys/kernel/random/boot_id | cut -c1-8) up=$(cut -d' ' -f1 /proc/uptime)"
