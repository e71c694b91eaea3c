#!/bin/bash
OUT=gpurun_out/r03zi; mkdir -p $OUT
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_driver_form_$i.json; done
python - <<'PY'
import json
for i in (1, 2, 3):
    d = json.loads(open(f"gpurun_out/r03zi/bench_driver_form_{i}.json").read())
    print(i, round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", d["roofline"].get("frac"), d.get("telemetry", {}).get("tick_wall_us"))
PY
timeout 400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_full.log
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu_full.log | tail -5
