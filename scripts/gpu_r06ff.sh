#!/bin/bash
# round 6 (debug): which literal of the specialised kernel makes the value-tag fuzz seeds 16 / 43 fail
out=gpurun_out/r06ff; mkdir -p $out
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
T="timeout 120 python -m pytest tests/test_fuzz_requests.py -q -m gpu -p no:cacheprovider -x"
K="value_tags_forced and (16-generic or 43-generic)"
run() { echo "== $1"; shift; env "$@" $T -k "$K" 2>&1 | tail -1 | cut -c1-120; }
ALL=",op_bits,n_ops,n_saves,n_steps,src_is_live,skip_live,dp_s,nt,cached_saves,live_rows,load_rows,live_pmask,nt_loads,mtab,vtags,save_rows,save_pmask,"
{
run "baseline" X=1
run "all kept, no unroll, no prefetch" GGRS_DBG_SPEC_KEEP=$ALL GGRS_DBG_SPEC_NO_UNROLL=1 GGRS_DBG_NO_PREFETCH=1
run "all kept, no unroll" GGRS_DBG_SPEC_KEEP=$ALL GGRS_DBG_SPEC_NO_UNROLL=1
run "all kept" GGRS_DBG_SPEC_KEEP=$ALL
run "no unroll" GGRS_DBG_SPEC_NO_UNROLL=1
for f in op_bits n_saves n_steps src_is_live skip_live dp_s nt cached_saves live_rows load_rows live_pmask nt_loads vtags save_rows save_pmask; do
  run "keep $f" GGRS_DBG_SPEC_KEEP=,$f,
done
run "spec off" GGRS_JIT_SPECIALISE_AFTER=0
} 2>&1 | tee $out/bisect.log
