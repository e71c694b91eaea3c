#!/bin/bash
out=gpurun_out/r06w; mkdir -p $out
B="timeout 600 python bench.py --sync --no-traffic --no-cpu-baseline --no-extra --steps 100"
$B > $out/bench_sync_selffold.json 2>> $out/bench.err; $B > $out/bench_sync_selffold_2.json 2>> $out/bench.err
GGRS_FOLD_FORWARD_MIN_WGS=1000000 $B > $out/bench_sync_finalize.json 2>> $out/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06w/bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f, round(j["value"] / 1e9, 1), "G ms", round(j["ms_per_step"], 4), "launch", round(r["avg_launch_us"], 1), r.get("other_kernels"))
    except Exception as e: print(f, "unreadable", e)
PY
rm -f gpurun_out/r06w/bench_sync_v*.json
timeout 600 python bench.py --no-extra --no-traffic --no-cpu-baseline > $out/bench_pipelined.json 2>> $out/bench.err
timeout 600 python bench.py --schema allhot --sync --no-traffic --no-cpu-baseline --no-extra > $out/bench_allhot_sync.json 2>> $out/bench.err
GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 600 python bench.py --schema allhot --sync --no-traffic --no-cpu-baseline --no-extra > $out/bench_allhot_sync_finalize.json 2>> $out/bench.err
timeout 600 python bench.py --entities 4000000 --sync --no-traffic --no-cpu-baseline --no-extra > $out/bench_4m_sync.json 2>> $out/bench.err
GGRS_FOLD_FORWARD_MIN_WGS=100000000 timeout 600 python bench.py --entities 4000000 --sync --no-traffic --no-cpu-baseline --no-extra > $out/bench_4m_sync_finalize.json 2>> $out/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06w/bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f, round(j["value"] / 1e9, 1), "G ms", round(j["ms_per_step"], 4), "launch", round(r["avg_launch_us"], 1), r.get("other_kernels"))
    except Exception as e: print(f, "unreadable", e)
PY
timeout 1200 python -m pytest tests/test_fuzz_requests.py tests/test_gpu_parity.py tests/test_gpu_gen_groups.py -q -m gpu -x 2>&1 | tail -2
