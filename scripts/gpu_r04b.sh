#!/bin/bash
# Round 4: A/B of the group fold (atomics-combine by wave 0) against GGRS_GROUP_FOLD_MIN_WGS=0 at 1 M / 2 M / 4 M, async + blocking.
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_parity.py tests/test_gpu_gen_groups.py tests/test_gpu_row_versions.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_subset.log
B="timeout 600 python bench.py --no-cpu-baseline"
for n in 1000000 2000000 4000000; do
  for rep in 1 2; do
    $B --entities $n --steps 20 --warmup 5 > $OUT/bench_drv_${n}_gf_$rep.json 2>> $OUT/bench.err
    GGRS_GROUP_FOLD_MIN_WGS=0 $B --entities $n --steps 20 --warmup 5 > $OUT/bench_drv_${n}_nogf_$rep.json 2>> $OUT/bench.err
  done
  $B --entities $n > $OUT/bench_long_${n}_gf.json 2>> $OUT/bench.err
  GGRS_GROUP_FOLD_MIN_WGS=0 $B --entities $n > $OUT/bench_long_${n}_nogf.json 2>> $OUT/bench.err
  $B --entities $n --sync > $OUT/bench_sync_${n}_gf.json 2>> $OUT/bench.err
  GGRS_GROUP_FOLD_MIN_WGS=0 $B --entities $n --sync > $OUT/bench_sync_${n}_nogf.json 2>> $OUT/bench.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04b"), "bench*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {}); t = (j.get("telemetry") or {}).get("tick_wall_us") or {}
        print(f"{os.path.basename(f):40s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  first-tick {t.get('first5', [0])[0]}  median {t.get('median')}  parity {j.get('parity', {}).get('synctest_resim_consistent_over_timed_ticks')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
