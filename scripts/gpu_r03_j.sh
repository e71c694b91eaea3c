#!/bin/bash
# round 3, call J: XCD-aware tile mapping of the generated kernel
O=gpurun_out/${1:-r03j}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_some.log 2>&1
tail -3 $O/pytest_some.log
run() { echo "== $1: $(env $2 timeout 120 benches/tick_bench $3 8 ${4:-200} 16 0 ${5:-0} 1 2>&1 | tail -n 1 | cut -c60-230)" | tee -a $O/plain.txt; }
for n in 100000 300000 1000000 4000000; do run "default n=$n" "A=1" $n; done
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python bench.py --no-cpu-baseline --no-checksum > $O/bench_nochecksum.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ("bench","bench_nochecksum"):
    j=json.load(open(f"gpurun_out/r03j/{f}.json")); r=j["roofline"]
    print(f, "value %.2fG ms/step %.4f kernel %.1f us frac %.3f" % (j["value"]/1e9, j["ms_per_step"], r["avg_launch_us"], r["frac"]), r.get("launch_us"))
PY
