#!/bin/bash
# Contiguous-arena failure, taken apart (scripts/contig_diag.py): the minimal sequence is a contiguous NO_GROUPS world, closed, then a
# PAGED (UNFUSED) world of the same size.  Arms run as successive processes on one fresh box.
OUT=gpurun_out/r03fd; mkdir -p $OUT; export TMPDIR=/tmp; rm -f $OUT/diag4.log
run() { tag=$1; seq=$2; shift 2; echo "=== $tag [$seq]: $*" >> $OUT/diag4.log; env DIAG_NO_READS=1 "$@" timeout 300 python scripts/contig_diag.py $seq 2>&1 | grep -vE "amdgpu.ids|one more|final state" | sed -e "s/, 'generated_kernel.*//" >> $OUT/diag4.log; }
run freed 8,2 GGRS_ARENA_CONTIG=2 GGRS_ARENA_PARK=0
run parked 8,2 GGRS_ARENA_CONTIG=2
run parked_longer 8,2,8,0,2,8:20000,2:20000,0:5000,2 GGRS_ARENA_CONTIG=2
run flagged_freed 40,2 GGRS_ARENA_PARK=0
run flagged_jit_freed 32,2 GGRS_ARENA_PARK=0
run flagged_parked 40,2,40,2 A=1
cat $OUT/diag4.log | cut -c1-200
for i in 1 2; do GGRS_ARENA_CONTIG=2 timeout 900 python -m pytest tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -n 2; done
