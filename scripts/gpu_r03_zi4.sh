#!/bin/bash
OUT=gpurun_out/r03zi; mkdir -p $OUT
timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc.json
timeout 200 python bench.py --config 4 --no-cpu-baseline --no-specialise-wait 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc_nowait.json
GGRS_JIT_SPECIALISE_AFTER=0 timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc_generic.json
timeout 200 python bench.py --config 2 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config2_gc.json
timeout 300 python bench.py --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_driver_form.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03zi/bench_config*_gc*.json")) + ["gpurun_out/r03zi/bench_driver_form.json"]:
    d = json.loads(open(f).read())
    print(f.split("/")[-1], round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", d["config"].get("specialised_kernel"), d.get("telemetry", {}).get("tick_wall_us"), d["parity"].get("equal", d["parity"]))
PY
