#!/bin/bash
# round 5, call f: full bench line (extra_configs), config 4 fold placement A/B, 2 M / 4 M with fold-forward, blocking API
TAG=${1:-r05f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_full.err | grep '^{' > $OUT/bench_full.json ) 2>&1 | tail -3
python - <<'PY'
import json,os
j=json.loads(open("gpurun_out/%s/bench_full.json" % os.environ.get("TAG","r05f")).read().strip().splitlines()[-1])
print("headline", round(j["value"]/1e9,1), j["ms_per_step"], j["parity"]["equal"], json.dumps(j["telemetry"]["host_timeline_us_per_tick"])[:300])
for k,v in j.get("extra_configs",{}).items():
    if isinstance(v,dict):
        print("==",k, v.get("seconds"), round(v.get("value",0)/1e9,2), v.get("ms_per_step"), (v.get("parity") or {}).get("equal"), (v.get("latency_floor") or {}).get("frac"), (v.get("roofline") or {}).get("frac"), (v.get("roofline_alu") or {}).get("frac"), v.get("error"), v.get("skipped"))
        c=v.get("c_loop")
        if c: print("   c_loop", c.get("ms_per_step"), c.get("kernel_us"), (c.get("latency_floor") or {}).get("frac"), (c.get("parity") or {}).get("equal"), json.dumps(c.get("two_in_flight"))[:300])
    else: print(k, v)
PY
tail -3 $OUT/bench_full.err
for v in 256 1024 4096; do GGRS_FOLD_FORWARD_MIN_WGS=$v timeout 300 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/err.txt | grep '^{' > $OUT/bench_config4_ff$v.json; done
for n in 300000 2000000 4000000; do timeout 300 python bench.py --entities $n --no-cpu-baseline --no-extra 2>> $OUT/err.txt | grep '^{' > $OUT/bench_$n.json; done
GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 300 python bench.py --entities 4000000 --no-cpu-baseline --no-extra 2>> $OUT/err.txt | grep '^{' > $OUT/bench_4000000_hostfold.json
timeout 300 python bench.py --sync --no-cpu-baseline 2>> $OUT/err.txt | grep '^{' > $OUT/bench_sync.json
timeout 300 python bench.py --no-cpu-baseline --no-extra 2>> $OUT/err.txt | grep '^{' > $OUT/bench_long.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % os.environ.get("TAG","r05f"))):
    if "full" in f: continue
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]; print(os.path.basename(f), round(j["value"]/1e9,1), round(j["ms_per_step"]*1e3,2), round(r["avg_launch_us"],2), round(r["frac"],3), (j.get("latency_floor") or {}).get("frac"), (j.get("parity") or {}).get("equal"), j.get("telemetry",{}).get("tick_wall_us",{}).get("median"))
    except Exception as e: print(f, "unreadable", e)
PY
tail -3 $OUT/err.txt
