#!/bin/bash
# round 6, call v: self-fold for blocking calls (the launch folds its own rows: no k_gen_finalize behind an HBM-sized blocking tick) -- suite, then --sync and the pipelined headline
out=gpurun_out/r06v; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -25 > $out/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $out/pytest_gpu.log | cut -c1-300
B="timeout 600 python bench.py"
$B --sync --no-traffic --cpu-ticks 1 > $out/bench_sync.json 2> $out/bench.err; echo "sync rc=$?"
$B --sync --no-traffic --no-cpu-baseline > $out/bench_sync_2.json 2>> $out/bench.err
$B --no-extra --no-traffic --no-cpu-baseline > $out/bench.json 2>> $out/bench.err
$B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extra > $out/bench_driver_form_short.json 2>> $out/bench.err
$B --schema allhot --sync --no-traffic --no-cpu-baseline > $out/bench_allhot_sync.json 2>> $out/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06v/bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f, round(j["value"] / 1e9, 1), "G ms", round(j["ms_per_step"], 4), "launch", round(r["avg_launch_us"], 1), "frac", round(r["frac"], 3), r.get("other_kernels"), (j.get("parity") or {}).get("equal"), (j.get("telemetry") or {}).get("tick_wall_us", {}).get("median"))
    except Exception as e: print(f, "unreadable", e)
PY
