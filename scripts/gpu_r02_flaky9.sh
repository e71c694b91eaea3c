#!/bin/bash
# first process on a fresh box: the suite up to the flaky case under a given environment
set -u
TAG=${1:-a}; shift
OUT=gpurun_out/r02fc; mkdir -p $OUT; export TMPDIR=/tmp
env "$@" timeout 900 python -m pytest tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/f9_$TAG.txt 2>&1
echo "f9 $TAG [$*]: $(grep -E 'passed|failed' $OUT/f9_$TAG.txt | tail -n 1) boot=$(cat /proc/sys/kernel/random/boot_id | cut -c1-8)"
