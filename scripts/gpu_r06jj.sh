#!/bin/bash
# round 6: FRESH fuzz seeds (500..539) for every HIP request-list fuzz under the defaults, then the generic value-tag / lazy-live worlds of those seeds with every shape specialised at first sight
out=gpurun_out/r06jj; mkdir -p $out
export GGRS_FUZZ_SEED0=500 GGRS_FUZZ_SEEDS=40
T="python -m pytest tests/test_fuzz_requests.py -q -m gpu -p no:cacheprovider"
echo "== fresh seeds, defaults"; timeout 400 $T -k "not every_shape_specialised" 2>&1 | tail -5 | cut -c1-300 | tee $out/fresh_defaults.log
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
echo "== fresh seeds, every shape specialised: value tags forced (generic worlds)"; timeout 330 $T -k "value_tags_forced and generic and not hbm and not every_shape" 2>&1 | tail -5 | cut -c1-300 | tee $out/fresh_tags_spec1.log
