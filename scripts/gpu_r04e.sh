#!/bin/bash
# Round 4, call E: fused spawn (suite; config 5 --spawn and a spawning SyncTest session with / without GGRS_JIT_FUSE_SPAWN), the bpr = 1 fan-out
# lines after the shared-prefix fix, the group fold's soak under REAL load (a second process streams the allhot world the whole time).
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
B="timeout 900 python bench.py"
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_spawn_fused.json
GGRS_JIT_FUSE_SPAWN=0 $B --config 5 --spawn --steps 4 --warmup 1 --preheat-ms 0 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_spawn_unfused.json
$B --fanout 2>> $OUT/bench.err | grep '^{' > $OUT/bench_fanout_ws1.json
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt 2>> $OUT/bench.err
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so $B --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_oversubscribed.json; echo "bench --gpus 2 rc=$?"
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29931 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_torchrun.json; echo "torchrun --gpus 2 rc=$?"
python scripts/spawn_session_bench.py > $OUT/spawn_session.txt 2>&1; tail -6 $OUT/spawn_session.txt
# ---- soak under real load
( timeout 300 python bench.py --schema allhot --steps 600000 --no-cpu-baseline --preheat-ms 0 > $OUT/soak_background_allhot.json 2>> $OUT/bench.err & )
sleep 15
$B --entities 4000000 --steps 400 --cpu-ticks 1 --parity-ticks 24 > $OUT/soak_4000000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak rc=$?"
GGRS_GROUP_FOLD_MIN_WGS=8 $B --entities 700000 --steps 600 --cpu-ticks 1 --parity-ticks 48 > $OUT/soak_700000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak2 rc=$?"
GGRS_GROUP_FOLD_MIN_WGS=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_row_versions.py -m gpu -x -q 2>&1 | tail -2 | tee $OUT/pytest_group_fold_forced_under_load.log
wait; sleep 1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04e"), "*.json"))):
    if not os.path.basename(f).startswith(("bench", "soak")): continue
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {}); a = j.get("roofline_alu") or {}
        print(f"{os.path.basename(f):48s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  n_gpus {j.get('n_gpus')}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
