#!/bin/bash
# round 6, call i: the first timed tick under rocprofv3 (--hip-trace --kernel-trace: no counters), and the driver's bench form with the new floor pricing
TAG=${1:-r06i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --hip-trace --kernel-trace -f csv -d $OUT/trace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
python scripts/first_tick_trace.py $OUT/trace $OUT/first_tick.json | head -80
find $OUT/trace -name '*.csv' -size +8M -delete; find $OUT -name '*.db' -delete
J() { grep '^{' ; }
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | J > $OUT/bench_driver_form.json; echo "bench rc=$?"; tail -3 $OUT/bench.err
TAG=$TAG python - <<'PY'
import json, os
j = json.loads(open(os.path.join("gpurun_out", os.environ["TAG"], "bench_driver_form.json")).read())
r = j["roofline"]
print("headline", round(j["value"]/1e9, 2), "G  ms/step", j["ms_per_step"], " launch us", r.get("avg_launch_us"), " frac", r.get("frac"), " parity", j["parity"]["equal"], "first5", j["telemetry"]["tick_wall_us"]["first5"])
for k, v in (j.get("extra_configs") or {}).items():
    if isinstance(v, dict) and "value" in v:
        lf = v.get("latency_floor") or {}
        print(" ", k, round(v["value"]/1e9, 2), "G ms/step", round(v["ms_per_step"], 5), "parity", (v.get("parity") or {}).get("equal"), "hbm", round((v.get("roofline") or {}).get("frac") or 0, 3), "alu", (v.get("roofline_alu") or {}).get("frac"), "floor", lf.get("frac"), lf.get("consistent"), "host", v.get("host_loop"), "py", ((v.get("python_driver") or {}).get("value") or 0)/1e9, v.get("floor_inconsistent"))
    else: print(" ", k, v)
PY
