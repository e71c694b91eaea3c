#!/bin/bash
# Contiguous-arena failure: which earlier tests the failing (paged, UNFUSED) world needs before it.
OUT=gpurun_out/r03fe; mkdir -p $OUT; export TMPDIR=/tmp
T='tests/test_gpu_parity.py::test_particles_synctest_checksums_and_state'
run() { tag=$1; shift; GGRS_ARENA_CONTIG=2 GGRS_DEBUG_ARENA=1 timeout 600 python -m pytest "$@" -m gpu -q -x -s > $OUT/$tag.txt 2>&1; echo "$tag: $(grep -E ' passed| failed' $OUT/$tag.txt | tail -n 1) $(grep -E '^FAILED' $OUT/$tag.txt | head -n 1 | cut -c1-120)" | tee -a $OUT/summary.txt; }
run a_this_test_all_params "$T"
run b_only_10000 "$T" -k "10000-8-30"
run c_only_flags2 "$T" -k "test_particles_synctest_checksums_and_state and 2-"
run d_parity_file tests/test_gpu_parity.py
run e_earlier_files_then_one tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py "$T[2-10000-8-30]"
run f_flags0_then_2 "$T[0-10000-8-30]" "$T[2-10000-8-30]"
run g_flags8_then_2 "$T[8-10000-8-30]" "$T[2-10000-8-30]"
run h_1025_then "$T[2-1025-7-24]" "$T[2-10000-8-30]"
