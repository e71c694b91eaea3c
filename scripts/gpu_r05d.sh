#!/bin/bash
# round 5, call d: tile-major fold-forward rows A/B + the new tests
TAG=${1:-r05d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_knobs.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
for rep in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_ff_$rep.json 2>> $OUT/err.txt
  GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_hostfold_$rep.json 2>> $OUT/err.txt
done
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_long_ff.json 2>> $OUT/err.txt
GGRS_FOLD_FORWARD_MIN_WGS=1000000 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_long_hostfold.json 2>> $OUT/err.txt
tail -5 $OUT/err.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % os.environ.get("TAG","r05d"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j["value"]/1e9,1), round(j["ms_per_step"]*1e3,2), round(j["roofline"]["avg_launch_us"],2), j["telemetry"]["tick_wall_us"]["first5"], j["telemetry"]["tick_wall_us"]["median"], j["telemetry"]["tick_wall_us"]["after_last_collect"])
    except Exception as e: print(f, "unreadable", e)
PY
