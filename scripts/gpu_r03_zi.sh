#!/bin/bash
# r03zi: one kernel per group shape (P2P sessions): the new parity test + the suites that drive the specialised kernel, then BASELINE config 4
# with and without waiting for the kernels
OUT=gpurun_out/r03zi; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_gen_groups.py tests/test_gpu_knobs.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5 > $OUT/pytest.log
cat $OUT/pytest.log
for i in 1 2; do
  timeout 200 python bench.py --config 4 --no-cpu-baseline --no-specialise-wait 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_nowait_$i.json
  timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03zi/bench_config4*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("/")[-1], round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", d["config"].get("specialised_kernel"), d["config"].get("specialise_settle"), d["parity"]["equal"])
    except Exception as e: print(f, "unreadable", e)
PY
tail -5 $OUT/bench.err
