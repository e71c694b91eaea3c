#!/bin/bash
# r03zj: GGRS_SPIN_WAIT_US -- poll before blocking.  The driver's 20-tick form, the blocking API and the small configs with 0 / 200 us; the GPU
# suite under the knob
OUT=gpurun_out/r03zj; mkdir -p $OUT
for s in 0 200; do
  GGRS_SPIN_WAIT_US=$s timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_driver_form_spin$s.json
  GGRS_SPIN_WAIT_US=$s timeout 60 python bench.py --sync --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_sync_spin$s.json
done
GGRS_SPIN_WAIT_US=200 timeout 60 python bench.py --config 2 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config2_spin200.json
GGRS_SPIN_WAIT_US=200 timeout 60 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_spin200.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03zj/bench_*.json")):
    try:
        d = json.loads(open(f).read())
        t = d.get("telemetry", {}).get("tick_wall_us", {})
        print(f.split("/")[-1], round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", "first5", t.get("first5"), "median", t.get("median"), "after", t.get("after_last_collect"))
    except Exception as e: print(f, "unreadable", e)
PY
GGRS_SPIN_WAIT_US=200 timeout 100 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu_spin200.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_spin200.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu_spin200.log | tail -3
