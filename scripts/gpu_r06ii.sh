#!/bin/bash
# round 6: the GPU files r06gg did not cover, with every group shape specialised at first sight (branch steps' member kernels, fan-out, spawns, custom systems, despawn markers, box_game)
out=gpurun_out/r06ii; mkdir -p $out
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
timeout 840 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_zfuzz_branches.py tests/test_gpu_zfanout.py tests/test_gpu_fused_spawn.py tests/test_gpu_custom_system.py tests/test_despawn_rollback.py tests/test_box_game.py tests/test_gpu_golden.py tests/test_hierarchy_links.py tests/test_rollback_ordered.py tests/test_gpu_ring.py tests/test_bench_fanout_parity.py tests/test_fanout_template.py 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -6 | cut -c1-300 | tee $out/rest2_spec1.log
