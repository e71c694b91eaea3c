#!/bin/bash
# round 6: 64-bit row versions (counter starting just below 2^32) -- the row-version / branch files and the request-list fuzz under the defaults, smoke
out=gpurun_out/r06mm; mkdir -p $out
timeout 215 python -m pytest -q -m gpu -p no:cacheprovider -x tests/test_gpu_row_versions.py tests/test_gpu_zfuzz_branches.py tests/test_fuzz_requests.py tests/test_gpu_parity.py -k "not every_shape and not hbm_sized" 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -5 | cut -c1-300 | tee $out/ver64.log
timeout 40 python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
