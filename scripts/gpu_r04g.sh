#!/bin/bash
# Round 4, call G: after the pre-heat fix (same step count on every rank) and the zero-copy spawn payloads: suite, 12 two-rank runs, config 5 --spawn.
TAG=${1:-r04g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=4 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt
DBL=$PWD/tests/cpp/_build/librccl_double.so
ok=0; bad=0
for i in 1 2 3 4 5 6; do
  for e in "A=1" "GGRS_JIT_LANE_FOLD=0"; do
    if env $e GGRS_RCCL_LIB=$DBL timeout 300 python bench.py --gpus 2 --oversubscribe --steps 6 --warmup 2 --preheat-ms 20 --entities 300000 --no-cpu-baseline > $OUT/run_$i.out 2> $OUT/run_$i.err; then ok=$((ok+1)); else bad=$((bad+1)); tail -3 $OUT/run_$i.err; fi
  done
done
echo "two-rank runs: ok=$ok bad=$bad" | tee $OUT/two_rank_runs.txt
B="timeout 900 python bench.py"
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_spawn_fused.json
python scripts/spawn_session_bench.py > $OUT/spawn_session.txt 2>&1; tail -3 $OUT/spawn_session.txt
GGRS_RCCL_LIB=$DBL $B --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_oversubscribed.json
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04g"), "bench*.json"))):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {})
    print(f"{os.path.basename(f):44s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  n_gpus {j.get('n_gpus')}  parity {(j.get('parity') or {}).get('equal')} preheat {j.get('preheat')}")
PY
