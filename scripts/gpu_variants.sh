#!/bin/bash
# Correctness soak of every non-default kernel variant: the parity + golden GPU tests under each A/B knob.
TAG=${1:-variants}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
T="tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_ring.py tests/test_despawn_rollback.py tests/test_gpu_gen_groups.py tests/test_box_game.py"
run() {
  echo "== $*" | tee -a $OUT/variants.log
  env "$@" timeout 300 python -m pytest $T -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -3 | tee -a $OUT/variants.log
}
run A=1
run GGRS_TICK_REST=0
run GGRS_TICK_VEC=1
run GGRS_TICK_VEC=41
run GGRS_TICK_VEC=4
run GGRS_TICK_VEC=4 GGRS_TICK_REST=0
run GGRS_TICK_GENERIC=1
run GGRS_TICK_NTLOAD=1 GGRS_TICK_LDS=40960
run GGRS_BLOCK_PAD=4096 GGRS_COL_PAD=256 GGRS_ARENA_ALIGN=2097152 GGRS_ARENA_SKEW=4096
