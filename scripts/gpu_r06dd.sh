#!/bin/bash
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
T="timeout 500 python -m pytest tests/test_fuzz_requests.py -q -m gpu -p no:cacheprovider"
echo "== tags forced, seeds 16 43 13"; $T -k "value_tags_forced and (16-generic or 43-generic or 13-generic)" 2>&1 | tail -4 | cut -c1-200
echo "== the same without the destination-tag prefetch"; GGRS_DBG_NO_PREFETCH=1 $T -k "value_tags_forced and (16-generic or 43-generic or 13-generic)" 2>&1 | tail -4 | cut -c1-200
echo "== no tags, lazy-live fuzz, generic"; $T -k "lazy_live_block_forced and generic" 2>&1 | tail -4 | cut -c1-200
