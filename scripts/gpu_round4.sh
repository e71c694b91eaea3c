#!/bin/bash
# One gpurun call (round 4): GPU parity tests, smoke, every bench line -- headline in the driver's exact form and the long form, the
# all-columns-hot schema, the reference's full POD schema, the blocking API, 2 M / 4 M, BASELINE configs 2 / 4 / 5 (+ its diverging-branches
# variant, fused and unfused), a spawning SyncTest session, the N = 2 line over the transport double in both launch forms, the full-copy mode -- rocprofv3 kernel-trace stats of the headline and allhot
# commands and their FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, as MI355X_MICROARCH.md prescribes), a soak of the on-chip
# group fold at 4 M beside a second process loading the GPU, tick_bench through the C ABI.
# Usage: gpurun -- 'bash scripts/gpu_round4.sh [tag]';   then   python scripts/collect_round.py <tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
B="timeout 600 python bench.py"
$B --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; echo "bench (driver form) rc=$?"; cut -c1-400 $OUT/bench_driver_form.json
for rep in 2 3; do $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$rep.json 2>> $OUT/bench.err; done
$B > $OUT/bench.json 2>> $OUT/bench.err; echo "bench rc=$?"
$B --schema allhot > $OUT/bench_allhot.json 2>> $OUT/bench.err
$B --schema full > $OUT/bench_schema_full.json 2>> $OUT/bench.err
$B --sync --no-cpu-baseline > $OUT/bench_sync.json 2>> $OUT/bench.err
GGRS_ROW_VERSIONS=0 $B --no-cpu-baseline > $OUT/bench_fullcopy.json 2>> $OUT/bench.err
$B --no-groups --no-cpu-baseline > $OUT/bench_nogroups.json 2>> $OUT/bench.err
for n in 100000 300000 600000 2000000; do $B --entities $n --no-cpu-baseline > $OUT/bench_$n.json 2>> $OUT/bench.err; done
$B --entities 4000000 --cpu-ticks 1 > $OUT/bench_4000000.json 2>> $OUT/bench.err
GGRS_GROUP_FOLD_MIN_WGS=0 $B --entities 4000000 --no-cpu-baseline > $OUT/bench_4000000_no_group_fold.json 2>> $OUT/bench.err
$B --config 2 > $OUT/bench_config2.json 2>> $OUT/bench.err
$B --config 4 > $OUT/bench_config4.json 2>> $OUT/bench.err
$B --config 5 --steps 20 --warmup 3 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu.json
$B --config 5 --steps 20 --warmup 3 --no-share-prefix --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu_per_branch_prefix.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu_spawn.json
GGRS_JIT_FUSE_SPAWN=0 $B --config 5 --spawn --steps 4 --warmup 1 --preheat-ms 0 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_1gpu_spawn_unfused.json
python scripts/spawn_session_bench.py > $OUT/spawn_session.txt 2>&1
$B --fanout 2>> $OUT/bench.err | grep '^{' > $OUT/bench_fanout_ws1.json
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt 2>> $OUT/bench.err
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so $B --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_oversubscribed.json; echo "bench --gpus 2 rc=$?"
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29931 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | grep '^{' > $OUT/bench_gpus2_torchrun.json; echo "torchrun --gpus 2 rc=$?"
# ---- soak of the on-chip group fold under UNEVEN load: a 4 M world (group fold on) with its in-run oracle parity gate over 24 ticks, while a
# second process streams the allhot world on the same GPU (the hand-off's failure modes only show under load, MI355X_MICROARCH.md)
( $B --schema allhot --steps 400000 --no-cpu-baseline --preheat-ms 0 > $OUT/soak_background_allhot.json 2>> $OUT/bench.err & )
sleep 12
$B --entities 4000000 --steps 400 --cpu-ticks 1 --parity-ticks 24 > $OUT/soak_4000000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak rc=$?"
GGRS_GROUP_FOLD_MIN_WGS=8 $B --entities 700000 --steps 600 --cpu-ticks 1 --parity-ticks 48 > $OUT/soak_700000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak2 rc=$?"
wait; sleep 2
./scripts/ubench_alu > $OUT/ubench_alu.txt 2>&1
g++ -O2 -std=c++17 -Iinclude benches/tick_bench.cpp -o benches/tick_bench -Lbevy_ggrs_amd -lggrs_hip -Wl,-rpath,'$ORIGIN/../bevy_ggrs_amd' 2>> $OUT/bench.err
for n in 10000 100000 1000000; do ./benches/tick_bench $n 8 200 16 0 0 1; done > $OUT/tick_bench_sizes.txt 2>&1
# ---- profiles: the DEFAULT command (pre-heat included): its last 200 tick-shaped launches are the timed region bench.py's own HIP events sample
BENCH="python bench.py --steps 100 --warmup 16 --no-cpu-baseline --preheat-ms 0"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats_allhot -o stats -- python bench.py --schema allhot --no-cpu-baseline > $OUT/prof_stats_allhot.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_allhot_fetch -o fetch -- $BENCH --schema allhot > $OUT/prof_allhot_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_allhot_write -o write -- $BENCH --schema allhot > $OUT/prof_allhot_write.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -f csv -d $OUT/prof_sq -o sq -- $BENCH > $OUT/prof_sq.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats_allhot $OUT/kernel_trace_steady_allhot.json > /dev/null 2>&1
find $OUT -name '*.db' -size +20M -delete
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04"), "*.json"))):
    if not os.path.basename(f).startswith(("bench", "soak")): continue
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        if "value" not in j: continue
        r = j.get("roofline", {}); a = j.get("roofline_alu") or {}; lf = j.get("latency_floor") or {}
        print(f"{os.path.basename(f):48s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  alu {a.get('frac')}  floor {lf.get('frac')}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
