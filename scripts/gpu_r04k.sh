#!/bin/bash
# Round 4, call K: the group fold with exchange reads -- suite, forced-group-fold suites, 4 M soak beside a loading process, A/B of the launch time.
TAG=${1:-r04k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.log
GGRS_GROUP_FOLD_MIN_WGS=8 GGRS_JIT_DP=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_row_versions.py tests/test_gpu_fused_spawn.py tests/test_gpu_gen_groups.py -m gpu -x -q 2>&1 | tail -2 | tee $OUT/pytest_group_fold_forced.log
B="timeout 900 python bench.py"
( timeout 240 python bench.py --schema allhot --steps 500000 --no-cpu-baseline --preheat-ms 0 > $OUT/soak_background_allhot.json 2>> $OUT/bench.err & )
sleep 15
$B --entities 4000000 --steps 1500 --cpu-ticks 1 --parity-ticks 60 > $OUT/soak_4000000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak rc=$?"
GGRS_GROUP_FOLD_MIN_WGS=8 $B --entities 1000000 --steps 3000 --cpu-ticks 1 --parity-ticks 100 > $OUT/soak_1000000_group_fold_under_load.json 2>> $OUT/bench.err; echo "soak2 rc=$?"
wait; sleep 1
$B --entities 4000000 --no-cpu-baseline > $OUT/bench_4000000.json 2>> $OUT/bench.err
GGRS_GROUP_FOLD_MIN_WGS=8 $B --no-cpu-baseline > $OUT/bench_1000000_group_fold_forced.json 2>> $OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04k"), "*.json"))):
    if not os.path.basename(f).startswith(("bench", "soak")): continue
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j.get("roofline", {}); p = j.get("parity") or {}
    print(f"{os.path.basename(f):48s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  parity {p.get('equal')} over {p.get('checked_saves')} Saves, resim-consistent {p.get('synctest_resim_consistent_over_timed_ticks')} over {p.get('timed_ticks')} ticks")
PY
