#!/bin/bash
# round 5, call e: fold-forward without the system-scope release (A/B against the host fold), full bench line with extra_configs
TAG=${1:-r05e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_ff_$rep.json 2>> $OUT/err.txt
  GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_hostfold_$rep.json 2>> $OUT/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % os.environ.get("TAG","r05e"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["value"]/1e9,1), round(j["ms_per_step"]*1e3,2), round(j["roofline"]["avg_launch_us"],2), j["telemetry"]["tick_wall_us"]["first5"], j["telemetry"]["tick_wall_us"]["median"], j["telemetry"]["tick_wall_us"]["after_last_collect"], j["config"].get("generated_kernel_origin"))
    except Exception as e: print(f, "unreadable", e)
PY
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
python - <<'PY'
import json,os
j=json.loads(open("gpurun_out/%s/bench_full.json" % os.environ.get("TAG","r05e")).read().strip().splitlines()[-1])
print("headline", round(j["value"]/1e9,1), j["ms_per_step"], j["parity"], json.dumps(j["telemetry"]["host_timeline_us_per_tick"]))
for k,v in j.get("extra_configs",{}).items(): print(k, json.dumps(v)[:900])
PY
tail -5 $OUT/bench_full.err
