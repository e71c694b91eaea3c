#!/bin/bash
# round 6, call e: value tags -- tests with the feature forced on, then the A/B per size (BENCH_VALUE_TAGS=0 / 1) of the worlds it is meant for
TAG=${1:-r06e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_row_versions.py tests/test_fuzz_requests.py -m gpu -x -q -k "value_tags or steady_state" --durations=5 -p no:cacheprovider > $OUT/pytest_vtags.log 2>&1; echo "pytest vtags rc=$?" | tee -a $OUT/pytest_vtags.log; tail -15 $OUT/pytest_vtags.log
B="timeout 600 python bench.py --no-extra --no-cpu-baseline"
J() { grep '^{' ; }
for vt in 0 1; do
  BENCH_VALUE_TAGS=$vt $B 2>> $OUT/bench.err | J > $OUT/bench_1m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --schema allhot 2>> $OUT/bench.err | J > $OUT/bench_allhot_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --entities 2000000 2>> $OUT/bench.err | J > $OUT/bench_2m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --entities 4000000 2>> $OUT/bench.err | J > $OUT/bench_4m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --schema full 2>> $OUT/bench.err | J > $OUT/bench_full_vt$vt.json
done
$B --entities 4000000 --cpu-ticks 1 2>> $OUT/bench.err | sed 's/--no-cpu-baseline//' | J > $OUT/bench_4m_default.json
tail -3 $OUT/bench.err
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"], "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(f"{os.path.basename(f):32s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):8.2f} us  bytes/launch {r.get('algorithmic_bytes_per_launch', 0)/1e6:8.1f} MB  frac {r.get('frac', 0):.3f}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
