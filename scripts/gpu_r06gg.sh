#!/bin/bash
# round 6: after the SCC clobber on set_lanes / store_lanes -- the value-tag fuzz (and the other request-list fuzzes) with EVERY group shape specialised at first sight
out=gpurun_out/r06gg; mkdir -p $out
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
T="timeout 1500 python -m pytest -q -m gpu -p no:cacheprovider"
echo "== tags forced, all seeds, every shape specialised"; $T tests/test_fuzz_requests.py -k "value_tags_forced" 2>&1 | tail -4 | cut -c1-200 | tee $out/fuzz_tags_spec1.log
echo "== the other fuzz files + parity files, every shape specialised"; $T tests/test_fuzz_requests.py tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_gpu_row_versions.py tests/test_gpu_round5.py tests/test_gpu_schema.py -k "not value_tags_forced and not lazy_live_block_forced" 2>&1 | tail -4 | cut -c1-200 | tee $out/rest_spec1.log
