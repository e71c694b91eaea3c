#!/bin/bash
# round 5, call a: launch-cost micro-benchmark + today's baseline of the round-4 code (driver form x2, C-ABI tick loop at 10 k / 100 k / 1 M)
TAG=${1:-r05a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
./scripts/ubench_launch/ubench_launch scripts/ubench_launch/kernels.hsaco > $OUT/ubench_launch.json 2> $OUT/ubench_launch.err; echo "ubench rc=$?"; cat $OUT/ubench_launch.json
g++ -O2 -std=c++17 -Iinclude benches/tick_bench.cpp -o benches/tick_bench -Lbevy_ggrs_amd -lggrs_hip -Wl,-rpath,'$ORIGIN/../bevy_ggrs_amd' 2>> $OUT/err.txt
for n in 10000 100000 1000000; do ./benches/tick_bench $n 8 2000 200 0 0 1; done > $OUT/tick_bench_sizes.txt 2>&1; cat $OUT/tick_bench_sizes.txt
for rep in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$rep.json 2>> $OUT/err.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05a/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_us"], j["telemetry"]["tick_wall_us"])
PY
