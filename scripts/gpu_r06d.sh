#!/bin/bash
# round 6, call d: full GPU suite after the fan-out rewrite + the 16-byte fold-forward publication; ff stress (10^6 ticks against a host-fold shadow); adopt A/B; the driver's bench form
TAG=${1:-r06d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log
timeout 900 python scripts/ff_stress.py 300000 1000000 > $OUT/ff_stress_300k.json 2> $OUT/ff_stress.err; echo "ff stress 300k rc=$?"; cat $OUT/ff_stress_300k.json
timeout 900 python scripts/ff_stress.py 1000000 300000 > $OUT/ff_stress_1m.json 2>> $OUT/ff_stress.err; echo "ff stress 1M rc=$?"; cat $OUT/ff_stress_1m.json
timeout 600 python scripts/adopt_ab.py 100000 7 > $OUT/adopt_ab_100k.json 2> $OUT/adopt_ab.err; cat $OUT/adopt_ab_100k.json
timeout 600 python scripts/adopt_ab.py 1000000 7 > $OUT/adopt_ab_1m.json 2>> $OUT/adopt_ab.err; cat $OUT/adopt_ab_1m.json
timeout 600 python scripts/adopt_ab.py 100000 2 > $OUT/adopt_ab_100k_k2.json 2>> $OUT/adopt_ab.err; cat $OUT/adopt_ab_100k_k2.json
J() { grep '^{' ; }
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | J > $OUT/bench_driver_form.json; echo "bench rc=$?"
python - <<'PY'
import json, os
j = json.loads(open(os.path.join("gpurun_out", os.environ.get("TAG", "r06d"), "bench_driver_form.json")).read())
r = j["roofline"]
print("headline", round(j["value"]/1e9, 2), "G  ms/step", j["ms_per_step"], " launch us", r.get("avg_launch_us"), " frac", r.get("frac"), " parity", j["parity"]["equal"])
for k, v in (j.get("extra_configs") or {}).items():
    if isinstance(v, dict) and "value" in v: print(" ", k, round(v["value"]/1e9, 2), "G ms/step", round(v["ms_per_step"], 4), "parity", (v.get("parity") or {}).get("equal"), "hbm", (v.get("roofline") or {}).get("frac"), "alu", (v.get("roofline_alu") or {}).get("frac"), "adopt", ((v.get("parity") or {}).get("adopt") or {}).get("equal_to_oracle_straight_line"))
    else: print(" ", k, v)
PY
