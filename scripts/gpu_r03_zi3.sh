#!/bin/bash
OUT=gpurun_out/r03zi; mkdir -p $OUT
{
GGRS_JIT_SPECIALISE_AFTER=4 DIAG_SETTLE=1 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=4 DIAG_SETTLE=1 DIAG_TORCH=1 timeout 100 python scripts/p2p_diag.py
} > $OUT/p2p_diag_async.txt 2>&1
cat $OUT/p2p_diag_async.txt | cut -c1-330
GGRS_JIT_SPECIALISE_SYNC=1 timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_syncbuild.json
timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_3.json
python - <<'PY'
import json
for f in ("bench_config4_syncbuild.json", "bench_config4_3.json"):
    d = json.loads(open("gpurun_out/r03zi/" + f).read())
    print(f, round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", d["config"].get("specialised_kernel"), d["config"].get("specialise_settle"), d["telemetry"])
PY
