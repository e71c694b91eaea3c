#!/bin/bash
# Round 4, call R: blocking calls poll k_gen_finalize's completion tags (GGRS_SPIN_WAIT_US) -- knob cases, A/B of the blocking API at 1 M / 2 M, then the suite.
TAG=${1:-r04r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp TAG
timeout 600 python -m pytest tests/test_gpu_knobs.py -m gpu -x -q -k "SPIN or HOST_FOLD or defaults" 2>&1 | grep -a "passed\|failed\|Error" | tail -3 | tee $OUT/pytest_knobs_spin.log
B="timeout 600 python bench.py --no-cpu-baseline"
for rep in 1 2; do
  $B --sync --steps 400 > $OUT/bench_sync_spin_$rep.json 2>> $OUT/bench.err
  GGRS_SPIN_WAIT_US=0 $B --sync --steps 400 > $OUT/bench_sync_stream_wait_$rep.json 2>> $OUT/bench.err
done
$B --sync --steps 300 --entities 2000000 > $OUT/bench_sync_2000000_spin.json 2>> $OUT/bench.err
GGRS_SPIN_WAIT_US=0 $B --sync --steps 300 --entities 2000000 > $OUT/bench_sync_2000000_stream_wait.json 2>> $OUT/bench.err
$B --sync --steps 400 --schema allhot > $OUT/bench_sync_allhot_spin.json 2>> $OUT/bench.err
GGRS_SPIN_WAIT_US=0 $B --sync --steps 400 --schema allhot > $OUT/bench_sync_allhot_stream_wait.json 2>> $OUT/bench.err
$B --config 2 > $OUT/bench_config2.json 2>> $OUT/bench.err
$B --config 2 > $OUT/bench_config2_b.json 2>> $OUT/bench.err
$B > $OUT/bench_default_long_form.json 2>> $OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04r"), "bench*.json"))):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r = j.get("roofline", {}); p = j.get("parity") or {}
    print(f"{os.path.basename(f):44s} {j['value']/1e9:8.2f} G  ms/step {j['ms_per_step']:.4f}  launch {r.get('avg_launch_us', 0):7.2f} us  frac {r.get('frac', 0):.3f}  floor {j.get('latency_floor', {}).get('frac')}  parity {p.get('equal')} over {p.get('checked_saves')}")
PY
tail -5 $OUT/bench.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error" | tail -3 | tee $OUT/pytest_gpu.log
