#!/bin/bash
# Round 4, call Z: the fan-out driver's patched request template with spawn payloads -- the fan-out tests, then config 5 with and without --spawn.
TAG=${1:-r04z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_zfanout.py tests/test_gpu_fused_spawn.py -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | cut -c1-500 | tail -6 | tee $OUT/pytest_fanout.log
timeout 60 python bench.py --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 --cpu-ticks 1 > $OUT/bench_config5_1gpu_spawn.json 2>> $OUT/bench.err; echo "rc=$?"
timeout 50 python bench.py --config 5 --steps 20 --warmup 3 --cpu-ticks 1 > $OUT/bench_config5_1gpu.json 2>> $OUT/bench.err; echo "rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04z/bench*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    p = j.get("parity") or {}
    print(f, round(j["value"] / 1e9, 1), "G", round(j["ms_per_step"], 3), "ms/step", "alu", (j.get("roofline_alu") or {}).get("frac"), "parity", p.get("equal"), p.get("checked_saves"))
PY
tail -3 $OUT/bench.err
