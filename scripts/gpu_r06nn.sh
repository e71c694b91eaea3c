#!/bin/bash
# round 6: 64-bit row versions -- files r06mm did not run, as many as the last two GPU-minutes hold
out=gpurun_out/r06nn; mkdir -p $out
timeout 120 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_schema.py tests/test_gpu_fused_spawn.py tests/test_despawn_rollback.py tests/test_box_game.py tests/test_gpu_custom_system.py tests/test_gpu_golden.py tests/test_gpu_zfanout.py tests/test_gpu_gen_groups.py 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -5 | cut -c1-300 | tee $out/ver64_rest.log
