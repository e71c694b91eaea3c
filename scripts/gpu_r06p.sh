#!/bin/bash
# round 6, call p: worlds beyond 4 M entities (the ring of a 32 M world is 21 GB of the 288): does the path hold, and at what fraction of peak
out=gpurun_out/r06p; mkdir -p $out
timeout 900 python bench.py --entities 8000000 --steps 10 --warmup 4 --preheat-ms 0 --parity-ticks 2 --cpu-ticks 1 --no-extra --no-traffic > $out/bench_8000000_parity.json 2> $out/bench_8000000_parity.err; echo "8 M with parity rc=$?"; tail -2 $out/bench_8000000_parity.err | cut -c1-300
for n in 8000000 16000000 32000000; do
  timeout 600 python bench.py --entities $n --steps 30 --warmup 20 --no-extra --no-traffic --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err; echo "$n rc=$?"; tail -2 $out/bench_$n.err | cut -c1-300
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06p/bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f, round(j["value"] / 1e9, 1), "G ms", round(j["ms_per_step"], 4), "launch", round(r["avg_launch_us"], 1), "frac", round(r["frac"], 3), "bytes", r["algorithmic_bytes_per_launch"], j.get("parity"))
    except Exception as e: print(f, "unreadable", e)
PY
