#!/bin/bash
# Round 4, call I: batched spawning branches -- suite, config 5 --spawn, the spawn session.
TAG=${1:-r04i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=4 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
B="timeout 900 python bench.py"
for rep in 1 2; do $B --config 5 --spawn --steps 12 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_spawn_$rep.json; done
$B --config 5 --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5.json
python scripts/spawn_session_bench.py > $OUT/spawn_session.txt 2>&1; tail -2 $OUT/spawn_session.txt
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04i"), "bench*.json"))):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {})
    print(f"{os.path.basename(f):40s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us x {r.get('launches_per_step')}  parity {(j.get('parity') or {}).get('equal')}")
PY
