#!/bin/bash
# Lazy live block: the new parity tests, the sessions that edit the world between lists (fuzzer, row versions, P2P), then the A/B at 1 M / 2 M / 4 M / allhot.
TAG=${1:-r05h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_gen_groups.py tests/test_gpu_fuzz.py tests/test_gpu_row_versions.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
B="timeout 600 python bench.py"
J() { grep '^{' ; }
for rep in 1 2 3; do
  $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>> $OUT/err.txt | J > $OUT/bench_driver_lazy_$rep.json
  $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-lazy-live 2>> $OUT/err.txt | J > $OUT/bench_driver_nolazy_$rep.json
done
$B --no-extra 2>> $OUT/err.txt | J > $OUT/bench_long_lazy.json
$B --no-extra --no-lazy-live --no-cpu-baseline 2>> $OUT/err.txt | J > $OUT/bench_long_nolazy.json
for n in 600000 2000000 4000000; do
  $B --entities $n --no-cpu-baseline 2>> $OUT/err.txt | J > $OUT/bench_${n}_lazy.json
  $B --entities $n --no-cpu-baseline --no-lazy-live 2>> $OUT/err.txt | J > $OUT/bench_${n}_nolazy.json
done
$B --schema allhot 2>> $OUT/err.txt | J > $OUT/bench_allhot_lazy.json
$B --schema allhot --no-lazy-live --no-cpu-baseline 2>> $OUT/err.txt | J > $OUT/bench_allhot_nolazy.json
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"], "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {})
        print(f"{os.path.basename(f):40s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  B/launch {r.get('algorithmic_bytes_per_launch')}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e: print(os.path.basename(f), "unreadable:", e)
PY
tail -5 $OUT/err.txt
