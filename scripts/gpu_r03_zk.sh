#!/bin/bash
# r03zk: sea_diffuse's middle step in 32-bit terms (device_prelude.hpp) -- GPU suite (every checksum against the oracle) + the headline bench
OUT=gpurun_out/r03zk; mkdir -p $OUT
timeout 100 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -3
timeout 40 python bench.py --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03zk/bench.json").read())
print(round(d["value"] / 1e9, 2), "G", round(d["ms_per_step"] * 1e3, 2), "us", "kernel", round(d["roofline"]["avg_launch_us"], 2), d["roofline"]["frac"], d["parity"].get("equal", d["parity"]))
PY
