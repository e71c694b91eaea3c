#!/bin/bash
# round 6: value-tag fuzz with every group shape specialised at once -- failure detail of the three seeds r06dd found
out=gpurun_out/r06ee; mkdir -p $out
export GGRS_JIT_SPECIALISE_AFTER=1 GGRS_JIT_SPECIALISE_SYNC=1
T="timeout 500 python -m pytest tests/test_fuzz_requests.py -q -m gpu -p no:cacheprovider -x"
for s in 13 16 43; do
  $T -k "value_tags_forced and ${s}-generic" 2>&1 | grep -v "^$" | tail -40 | cut -c1-600 > $out/seed$s.log
done
echo "== particles, tags forced"; $T -k "value_tags_forced and particles" 2>&1 | tail -4 | cut -c1-300 | tee $out/particles.log
echo "== AFTER=2"; GGRS_JIT_SPECIALISE_AFTER=2 $T -k "value_tags_forced and (16-generic or 43-generic or 13-generic)" 2>&1 | tail -4 | cut -c1-200 | tee $out/after2.log
tail -25 $out/seed13.log
