#!/bin/bash
# The contiguous-arena hazard (profiles/r02fc), one trial = the FIRST process on a fresh box:
#   arm mixed        GGRS_ARENA_CONTIG=2                      particles worlds contiguous (uncached), every other world paged (cached)
#   arm mixed_flush  GGRS_ARENA_CONTIG=2 GGRS_ARENA_FLUSH=1   the same + a system-scope L2 write-back / invalidate on every XCD before a contiguous arena is first used
#   arm mixed_freed  GGRS_ARENA_CONTIG=2 GGRS_ARENA_PARK=0    what "mixed" was before contiguous arenas were parked instead of freed
#   arm all_contig   GGRS_ARENA_CONTIG=1                      every world contiguous (round 2: 4 of 10 fresh boxes failed), arenas parked
# usage: gpurun -- 'bash scripts/gpu_r03_fc.sh <arm> <trial>'
ARM=$1; TRIAL=$2; OUT=gpurun_out/r03fc; mkdir -p $OUT; export TMPDIR=/tmp
case $ARM in mixed) E="GGRS_ARENA_CONTIG=2";; mixed_flush) E="GGRS_ARENA_CONTIG=2 GGRS_ARENA_FLUSH=1";; mixed_freed) E="GGRS_ARENA_CONTIG=2 GGRS_ARENA_PARK=0";; all_contig) E="GGRS_ARENA_CONTIG=1";; *) E="A=1";; esac
env $E timeout 900 python -m pytest tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/${ARM}_$TRIAL.txt 2>&1
echo "$ARM $TRIAL [$E]: $(grep -E 'passed|failed' $OUT/${ARM}_$TRIAL.txt | tail -n 1) $(grep -E '^FAILED' $OUT/${ARM}_$TRIAL.txt | head -n 1 | cut -c1-160) boot=$(cut -c1-8 /proc/sys/kernel/random/boot_id)" | tee $OUT/${ARM}_$TRIAL.summary
