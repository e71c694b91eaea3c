#!/bin/bash
# round 6, call o: the bench after the floor presentation change and with stdout held to the one JSON line (driver form with extras; small sizes); the bench tests
out=gpurun_out/r06o; mkdir -p $out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench.err; echo "driver form rc=$? lines on stdout: $(wc -l < $out/bench_driver_form.json)"
for n in 100000 300000; do timeout 300 python bench.py --entities $n --no-extra --no-traffic > $out/bench_$n.json 2>> $out/bench.err; echo "$n rc=$?"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06o/bench*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(j["value"] / 1e9, 1), j.get("floor_inconsistent"), json.dumps({k: v for k, v in (j.get("latency_floor") or {}).items() if k.startswith("frac") and k != "frac_basis" or k == "consistent"}))
    for k, v in (j.get("extra_configs") or {}).items():
        if isinstance(v, dict): print("   ", k, round(v.get("value", 0) / 1e9, 1), (v.get("parity") or {}).get("equal"), v.get("floor_inconsistent"), json.dumps({a: b for a, b in (v.get("latency_floor") or {}).items() if a.startswith("frac") and a != "frac_basis" or a == "consistent"}))
PY
timeout 900 python -m pytest tests/test_gpu_zfanout.py -q -m gpu -k "bench" 2>&1 | tail -3
