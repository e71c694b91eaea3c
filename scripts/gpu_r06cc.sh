#!/bin/bash
# round 6: the request-list fuzz and the parity / group files with every group's rows folded forward or by the launch itself (GGRS_FOLD_FORWARD_MIN_WGS=0), then by default
out=gpurun_out/r06cc; mkdir -p $out
GGRS_FOLD_FORWARD_MIN_WGS=0 timeout 1500 python -m pytest tests/test_fuzz_requests.py tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_gpu_row_versions.py tests/test_gpu_fused_spawn.py tests/test_gpu_round5.py tests/test_gpu_custom_system.py tests/test_gpu_schema.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $out/pytest_ff0.log | cut -c1-400
