#!/bin/bash
# One gpurun call (round 6): the whole GPU suite, smoke, the driver's bench command (headline + measured traffic + extra_configs), the long form, schemas, sizes (value tags
# by policy from 3 M), the blocking API, full-copy mode, the per-request path, config 5 in every form (compact / request list, retain, spawns), the N = 2 line over the transport
# double, adoption A/B, fold-forward stress, micro-benchmarks, a device-spawn session, rocprofv3 kernel-trace stats of the headline / allhot / config-5-retain commands and their
# FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, counters only: MI355X_MICROARCH.md).
# Usage: gpurun -- 'bash scripts/gpu_round6.sh [tag]';   then   python scripts/collect_round.py <tag>
TAG=${1:-r06z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname > $OUT/rocm_smi.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
fi
B="timeout 900 python bench.py"
J() { grep '^{' ; }
$B --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | J > $OUT/bench_driver_form.json; echo "bench (driver form, measured traffic, extra_configs) rc=$?"; cut -c1-300 $OUT/bench_driver_form.json
for rep in 2 3; do $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_driver_form_$rep.json; done
$B --no-extra --no-traffic 2>> $OUT/bench.err | J > $OUT/bench.json; echo "bench rc=$?"
$B --schema allhot --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_allhot.json
$B --schema full --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_schema_full.json
$B --sync --no-cpu-baseline --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_sync.json
GGRS_ROW_VERSIONS=0 $B --no-cpu-baseline --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_fullcopy.json
$B --no-groups --no-cpu-baseline --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_nogroups.json
GGRS_NO_HIPRTC=1 $B --no-cpu-baseline --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_no_hiprtc_shipped_objects.json
for n in 100000 300000 600000 2000000 3000000; do $B --entities $n --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_$n.json; done
$B --entities 4000000 --cpu-ticks 1 2>> $OUT/bench.err | J > $OUT/bench_4000000.json
BENCH_VALUE_TAGS=0 $B --entities 4000000 --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_4000000_no_value_tags.json
$B --schema allhot --entities 2000000 --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_allhot_2000000.json
$B --entities 16000000 --steps 30 --warmup 20 --no-cpu-baseline --no-extra --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_16000000.json
$B --config 2 2>> $OUT/bench.err | J > $OUT/bench_config2.json
$B --config 4 2>> $OUT/bench.err | J > $OUT/bench_config4.json
$B --config 5 --steps 20 --warmup 3 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu.json
$B --config 5 --steps 20 --warmup 3 --no-compact 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu_request_list.json
$B --config 5 --steps 20 --warmup 3 --retain all 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu_retain_all.json
$B --config 5 --steps 20 --warmup 3 --retain newest 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu_retain_newest.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu_spawn.json
$B --config 5 --spawn --steps 8 --warmup 2 --preheat-ms 0 --retain all 2>> $OUT/bench.err | J > $OUT/bench_config5_1gpu_spawn_retain_all.json
python scripts/spawn_session_bench.py > $OUT/spawn_session.txt 2>&1
timeout 300 python scripts/device_spawn_bench.py 70000 95000 > $OUT/device_spawn_session.txt 2>&1
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt 2>> $OUT/bench.err
GGRS_RCCL_LIB=$PWD/tests/cpp/_build/librccl_double.so $B --gpus 2 --oversubscribe --steps 20 --warmup 5 2>> $OUT/bench.err | J > $OUT/bench_gpus2_oversubscribed.json; echo "bench --gpus 2 rc=$?"
timeout 600 python scripts/adopt_ab.py 100000 7 > $OUT/adopt_ab_100k.json 2>> $OUT/bench.err
timeout 600 python scripts/adopt_ab.py 1000000 7 > $OUT/adopt_ab_1m.json 2>> $OUT/bench.err
timeout 600 python scripts/ff_stress.py 300000 1000000 > $OUT/ff_stress_300k.json 2>> $OUT/bench.err
timeout 120 ./scripts/ubench_launch/ubench_launch scripts/ubench_launch/kernels.hsaco > $OUT/ubench_launch.json 2>&1
[ -x scripts/ubench_alu ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench_alu.hip -o scripts/ubench_alu 2>> $OUT/bench.err
./scripts/ubench_alu > $OUT/ubench_alu.txt 2>&1
# ---- profiles: the DEFAULT command's timed launches (kernel trace), then the counter passes on a shorter form of it
BENCH="python bench.py --steps 100 --warmup 16 --no-cpu-baseline --preheat-ms 0 --no-extra --no-traffic"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-extra --no-traffic > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats_allhot -o stats -- python bench.py --schema allhot --no-cpu-baseline --no-traffic > $OUT/prof_stats_allhot.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats_config5_retain -o stats -- python bench.py --config 5 --retain all --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_stats_config5_retain.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/prof_allhot_fetch -o fetch -- $BENCH --schema allhot > $OUT/prof_allhot_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/prof_allhot_write -o write -- $BENCH --schema allhot > $OUT/prof_allhot_write.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -f csv -d $OUT/prof_sq -o sq -- $BENCH > $OUT/prof_sq.log 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats $OUT/kernel_trace_steady.json > /dev/null 2>&1
python scripts/kernel_trace_steady.py $OUT/prof_stats_allhot $OUT/kernel_trace_steady_allhot.json > /dev/null 2>&1
f=$(find $OUT/prof_stats_config5_retain -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_config5_retain.csv
find $OUT -name '*.db' -size +20M -delete; find $OUT -name '*kernel_trace.csv' -size +8M -delete
TAG=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r06z"), "*.json"))):
    if not os.path.basename(f).startswith(("bench",)): continue
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        if "value" not in j: continue
        r = j.get("roofline", {}); a = j.get("roofline_alu") or {}; lf = j.get("latency_floor") or {}
        print(f"{os.path.basename(f):52s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  alu {a.get('frac')}  floor {lf.get('frac')}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
