#!/bin/bash
# round 6, call h: the whole GPU suite (fan-out rewrite, 16-byte fold-forward cells, value tags), then the value-tag A/B at the sizes the policy turns it on for
TAG=${1:-r06h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log
B="timeout 600 python bench.py --no-extra --no-cpu-baseline"
J() { grep '^{' ; }
for vt in 0 1; do
  BENCH_VALUE_TAGS=$vt $B --entities 4000000 2>> $OUT/bench.err | J > $OUT/bench_4m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --entities 3000000 2>> $OUT/bench.err | J > $OUT/bench_3m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --schema allhot --entities 2000000 2>> $OUT/bench.err | J > $OUT/bench_allhot2m_vt$vt.json
  BENCH_VALUE_TAGS=$vt $B --schema allhot 2>> $OUT/bench.err | J > $OUT/bench_allhot_vt$vt.json
done
timeout 900 python bench.py --entities 4000000 --cpu-ticks 1 --no-extra 2>> $OUT/bench.err | J > $OUT/bench_4m_default_parity.json
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"], "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(f"{os.path.basename(f):36s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):8.2f} us  bytes/launch {r.get('algorithmic_bytes_per_launch', 0)/1e6:8.1f} MB  frac {r.get('frac', 0):.3f}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
