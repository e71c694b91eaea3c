#!/bin/bash
# Experiment: what would skipping the Saves' rows whose VALUES never change buy (translation.z, velocity.x, velocity.z: columns 2, 10, 12)?  A test hook drops them
# from every Save of a long-running steady session (the slots hold the same bytes already, so parity stays true); plus the bench after the event pool.
TAG=${1:-r05k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="timeout 600 python bench.py"
J() { grep '^{' ; }
$B --gpus 1 --steps 20 --warmup 5 2>> $OUT/err.txt | J > $OUT/bench_driver_form.json
for n in 1000000 2000000 4000000; do
  $B --entities $n --no-cpu-baseline --no-extra 2>> $OUT/err.txt | J > $OUT/bench_${n}_base.json
  BENCH_DBG_SKIP_ROWS=0x1404 $B --entities $n --no-cpu-baseline --no-extra 2>> $OUT/err.txt | J > $OUT/bench_${n}_skip3rows.json
done
$B --schema allhot --no-cpu-baseline 2>> $OUT/err.txt | J > $OUT/bench_allhot_base.json
BENCH_DBG_SKIP_ROWS=0x1404 $B --schema allhot --no-cpu-baseline 2>> $OUT/err.txt | J > $OUT/bench_allhot_skip3rows.json
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"], "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {})
        print(f"{os.path.basename(f):40s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  B/launch {r.get('algorithmic_bytes_per_launch')}  parity {(j.get('parity') or {}).get('equal')}")
        for k, v in (j.get("extra_configs") or {}).items():
            if isinstance(v, dict) and "c_loop" in v: print("   ", k, "c_loop", v["c_loop"].get("ms_per_step"), v["c_loop"].get("kernel_us"), (v["c_loop"].get("latency_floor") or {}).get("frac"), v["c_loop"].get("platform_floor"))
    except Exception as e: print(os.path.basename(f), "unreadable:", e)
PY
tail -3 $OUT/err.txt
