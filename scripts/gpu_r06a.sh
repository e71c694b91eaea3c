#!/bin/bash
# round 6, call a: the fan-out rewrite (member records, compact steps, retained branch states, adoption) -- its tests, then config 5 in every form
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_zfanout.py -x -q --durations=8 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_fanout.log; echo "pytest fanout rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_fanout.log; tail -30 $OUT/pytest_fanout.log
B="timeout 600 python bench.py"
J() { grep '^{' ; }
$B --config 5 --steps 20 --warmup 3 2> $OUT/bench.err | J > $OUT/bench_config5.json
$B --config 5 --steps 20 --warmup 3 --no-compact 2>> $OUT/bench.err | J > $OUT/bench_config5_list.json
$B --config 5 --steps 20 --warmup 3 --retain all 2>> $OUT/bench.err | J > $OUT/bench_config5_retain.json
$B --config 5 --steps 20 --warmup 3 --retain newest 2>> $OUT/bench.err | J > $OUT/bench_config5_retain_newest.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | J > $OUT/bench_config5_spawn.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 --no-compact 2>> $OUT/bench.err | J > $OUT/bench_config5_spawn_list.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 --retain all 2>> $OUT/bench.err | J > $OUT/bench_config5_spawn_retain.json
tail -5 $OUT/bench.err
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"], "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {}); a = j.get("roofline_alu") or {}
        print(f"{os.path.basename(f):44s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):8.2f} us x {r.get('launches_per_step')}  hbm {r.get('frac', 0):.3f}  alu {a.get('frac')}  parity {(j.get('parity') or {}).get('equal')} adopt {((j.get('parity') or {}).get('adopt') or {}).get('equal_to_oracle_straight_line')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
