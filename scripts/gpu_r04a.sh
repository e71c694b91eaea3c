#!/bin/bash
# (historic: the two full-copy lines below drove GGRS_TICK_GENERIC / k_tick3, which the commit after this run removed)
# Round 4, first GPU call: the suite on the new code (group fold, allhot schema, bench.py --gpus 2 over the transport double), then
# A/B of the group fold (GGRS_GROUP_FOLD_MIN_WGS=0 = round-3 behaviour) in the async and blocking host APIs, the all-columns-hot world,
# the N = 2 line at 1 M, 2 M / 4 M.   Usage: gpurun -- 'bash scripts/gpu_r04a.sh [tag]'
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
B="timeout 600 python bench.py"
$B --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; echo "bench (driver form) rc=$?"; cut -c1-400 $OUT/bench_driver_form.json
GGRS_GROUP_FOLD_MIN_WGS=0 $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_nogf.json 2>> $OUT/bench.err
$B --no-cpu-baseline > $OUT/bench_long.json 2>> $OUT/bench.err
GGRS_GROUP_FOLD_MIN_WGS=0 $B --no-cpu-baseline > $OUT/bench_long_nogf.json 2>> $OUT/bench.err
$B --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err
GGRS_GROUP_FOLD_MIN_WGS=0 $B --sync --no-cpu-baseline > $OUT/bench_sync_nogf.json 2>> $OUT/bench.err
$B --schema allhot > $OUT/bench_allhot.json 2>> $OUT/bench.err
GGRS_ROW_VERSIONS=0 GGRS_TICK_GENERIC=1 $B --no-cpu-baseline > $OUT/bench_fullcopy_generated.json 2>> $OUT/bench.err
GGRS_ROW_VERSIONS=0 $B --no-cpu-baseline > $OUT/bench_fullcopy_tick3.json 2>> $OUT/bench.err
for n in 2000000 4000000; do $B --entities $n --no-cpu-baseline > $OUT/bench_$n.json 2>> $OUT/bench.err; $B --entities $n --schema allhot --no-cpu-baseline > $OUT/bench_allhot_$n.json 2>> $OUT/bench.err; done
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt 2>> $OUT/bench.err
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so $B --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_oversubscribed.json; echo "bench --gpus 2 rc=$?"
$B --config 5 --steps 20 --warmup 3 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu.json
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats_allhot -o stats -- python bench.py --schema allhot --no-cpu-baseline > $OUT/prof_stats_allhot.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats_allhot $OUT/kernel_trace_steady_allhot.json > /dev/null 2>&1
find $OUT -name '*.db' -size +20M -delete
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04a"), "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(f"{os.path.basename(f):42s} value {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  B/ent {r.get('algorithmic_bytes_per_entity', 0):.0f}  parity {j.get('parity', {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
