#!/bin/bash
# Round 4, call U: the request-list fuzzer (tests/test_fuzz_requests.py) under every kernel-selecting knob, and on fresh seeds under the defaults.
TAG=${1:-r04u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F="timeout 300 python -m pytest tests/test_fuzz_requests.py -m gpu -q -x -k random_request_lists"
run() { name=$1; shift; echo "== $name"; env "$@" $F 2>&1 | grep -a "passed\|failed\|Error\|seed " | cut -c1-600 | tail -8 | tee -a $OUT/fuzz_$name.log; }
export GGRS_FUZZ_SEEDS=70
run tick_jit_0 GGRS_TICK_JIT=0
run persistent GGRS_JIT_PERSIST_MIN_SLOTS=1
run group_fold GGRS_GROUP_FOLD_MIN_WGS=8 GGRS_JIT_DP=0
run group_fold_device GGRS_GROUP_FOLD_MIN_WGS=8 GGRS_JIT_DP=0 GGRS_HOST_FOLD_MAX_WGS=0
run finalize GGRS_HOST_FOLD_MAX_WGS=0
run specialise_after_2 GGRS_JIT_SPECIALISE_AFTER=2 GGRS_JIT_SPECIALISE_SYNC=1     # (AFTER=1 compiles a kernel for every list of a random session: minutes, run r04u timed out on it)
run no_row_versions GGRS_ROW_VERSIONS=0
run no_fused_spawn GGRS_JIT_FUSE_SPAWN=0
run no_dead_groups GGRS_DEAD_GROUPS=0
run dp2 GGRS_JIT_DP=2
run contig GGRS_ARENA_CONTIG=1
export GGRS_FUZZ_SEEDS=400 GGRS_FUZZ_SEED0=5000
run fresh_seeds GGRS_X=0
