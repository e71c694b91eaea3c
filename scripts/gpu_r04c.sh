#!/bin/bash
# Round 4, call C: the suite after k_tick3's removal / shared fan-out prefix / token-aware specialiser; config 5 A/B of the shared prefix;
# first-Save cache limit sweep at 2 M / 4 M (headline and allhot schemas); small-world repeatability (configs 2, 4, 100 k: three fresh
# processes each) and a hip + kernel trace of one config-2 run.
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log
B="timeout 600 python bench.py"
./scripts/ubench_alu > $OUT/ubench_alu.txt 2>&1 || (cd scripts && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ubench_alu.hip -o ubench_alu && ./ubench_alu > ../$OUT/ubench_alu.txt 2>&1)
for rep in 1 2; do
  $B --config 5 --steps 20 --warmup 3 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_shared_$rep.json
  $B --config 5 --steps 20 --warmup 3 --no-share-prefix --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config5_perbranch_$rep.json
done
for mb in 80 160 320; do
  for n in 2000000 4000000; do
    GGRS_JIT_CACHED_SAVE_MAX_MB=$mb $B --entities $n --no-cpu-baseline > $OUT/bench_${n}_cache${mb}.json 2>> $OUT/bench.err
    GGRS_JIT_CACHED_SAVE_MAX_MB=$mb $B --entities $n --schema allhot --no-cpu-baseline > $OUT/bench_allhot_${n}_cache${mb}.json 2>> $OUT/bench.err
  done
done
GGRS_JIT_CACHED_SAVE_MAX_MB=160 $B --schema allhot --no-cpu-baseline > $OUT/bench_allhot_1000000_cache160.json 2>> $OUT/bench.err
for rep in 1 2 3; do
  $B --config 2 --no-cpu-baseline > $OUT/bench_config2_$rep.json 2>> $OUT/bench.err
  $B --config 4 --no-cpu-baseline > $OUT/bench_config4_$rep.json 2>> $OUT/bench.err
  $B --entities 100000 --no-cpu-baseline > $OUT/bench_100000_$rep.json 2>> $OUT/bench.err
done
$B --config 2 > $OUT/bench_config2_full.json 2>> $OUT/bench.err
$B --config 4 > $OUT/bench_config4_full.json 2>> $OUT/bench.err
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -f csv -d $OUT/prof_config2 -o c2 -- python bench.py --config 2 --no-cpu-baseline --steps 200 > $OUT/prof_config2.log 2>&1
find $OUT -name '*.db' -size +20M -delete; find $OUT -name '*hip_api_trace.csv' -size +30M -delete
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04c"), "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {}); t = (j.get("telemetry") or {}).get("tick_wall_us") or {}; a = j.get("roofline_alu") or {}; lf = j.get("latency_floor") or {}
        print(f"{os.path.basename(f):44s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  alu {a.get('frac')}  floor {lf.get('frac')}  median {t.get('median')}  parity {(j.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
