#!/bin/bash
# Round 4, call S: soak of the polled blocking wait (does the process grow when the stream is hardly ever waited for?), the driver's form again.
TAG=${1:-r04s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp TAG
B="timeout 600 python bench.py"
$B --sync --entities 300000 --steps 150000 --cpu-ticks 1 --parity-ticks 60 > $OUT/soak_sync_300000.json 2>> $OUT/bench.err; echo "rc=$?"
$B --sync --steps 60000 --cpu-ticks 1 --parity-ticks 60 > $OUT/soak_sync_1000000.json 2>> $OUT/bench.err; echo "rc=$?"
GGRS_SPIN_WAIT_US=0 $B --sync --entities 300000 --steps 150000 --cpu-ticks 1 --parity-ticks 60 > $OUT/soak_sync_300000_stream_wait.json 2>> $OUT/bench.err; echo "rc=$?"
for rep in 1 2 3; do $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$rep.json 2>> $OUT/bench.err; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04s"), "*.json"))):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r = j.get("roofline", {}); p = j.get("parity") or {}; t = j.get("telemetry") or {}
    print(f"{os.path.basename(f):44s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  rss {t.get('rss_mb')}  parity {p.get('equal')} over {p.get('checked_saves')}, resim-consistent {p.get('synctest_resim_consistent_over_timed_ticks')} over {p.get('timed_ticks')}  wall {(t.get('tick_wall_us') or {}).get('first5')} .. after {(t.get('tick_wall_us') or {}).get('after_last_collect')}")
PY
tail -3 $OUT/bench.err
