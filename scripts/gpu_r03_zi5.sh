#!/bin/bash
OUT=gpurun_out/r03zi; mkdir -p $OUT
timeout 300 python bench.py --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_driver_form.json
timeout 300 python bench.py 2>> $OUT/bench.err | grep '^{' > $OUT/bench.json
timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4.json
timeout 200 python bench.py --config 2 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config2.json
timeout 200 python bench.py --entities 100000 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_100000.json
python - <<'PY'
import json
for f in ("bench_driver_form", "bench", "bench_config4", "bench_config2", "bench_100000"):
    d = json.loads(open(f"gpurun_out/r03zi/{f}.json").read())
    print(f, round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", d["roofline"].get("frac"), d["config"].get("specialised_kernel"), d.get("telemetry", {}).get("tick_wall_us"), d["parity"].get("equal", d["parity"]))
PY
timeout 400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
