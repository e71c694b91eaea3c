#!/bin/bash
# round 5, call b: first GPU validation of the refactor (fold-forward, per-world argument block, ABI v8): GPU suite, smoke, bench in the driver's form, C-ABI tick loop
TAG=${1:-r05c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?"; tail -30 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
for n in 10000 100000 1000000; do ./benches/tick_bench $n 8 2000 200 0 0 1; done > $OUT/tick_bench_sizes.txt 2>&1; cat $OUT/tick_bench_sizes.txt
for rep in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$rep.json 2>> $OUT/err.txt; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form_3.json 2>> $OUT/err.txt
GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_hostfold.json 2>> $OUT/err.txt
tail -5 $OUT/err.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % os.environ.get("TAG","r05c"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["value"]/1e9,1), round(j["ms_per_step"]*1e3,2), round(j["roofline"]["avg_launch_us"],2), j["telemetry"]["tick_wall_us"], (j.get("parity") or {}).get("equal"))
    except Exception as e: print(f, "unreadable", e)
PY
