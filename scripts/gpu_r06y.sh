#!/bin/bash
# round 6, call y: blocking calls on very large worlds (self-fold only while its waiting workgroups are few: 8 M and beyond keep k_gen_finalize)
out=gpurun_out/r06y; mkdir -p $out
for n in 8000000 32000000; do timeout 600 python bench.py --entities $n --sync --steps 20 --warmup 20 --no-extra --no-traffic --no-cpu-baseline > $out/bench_sync_$n.json 2> $out/bench_$n.err; echo "$n rc=$?"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06y/bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f, round(j["value"] / 1e9, 1), "G ms", round(j["ms_per_step"], 4), "launch", round(r["avg_launch_us"], 1), r.get("other_kernels"), j.get("parity"))
    except Exception as e: print(f, "unreadable", e)
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "blocking_calls_fold" 2>&1 | tail -1
