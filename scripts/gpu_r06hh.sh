#!/bin/bash
# round 6, last tree (after the SCC clobber): the whole GPU suite, smoke, the driver's bench command, the value-tag sizes (their kernels are the ones whose schedule the
# clobber changes), rocprofv3 kernel-trace stats of the default bench command
OUT=gpurun_out/r06hh; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' > $OUT/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
B="timeout 600 python bench.py"
J() { grep '^{' ; }
$B --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | J > $OUT/bench_driver_form.json; echo "bench (driver form) rc=$?"; cut -c1-400 $OUT/bench_driver_form.json
$B --entities 3000000 --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_3000000.json
$B --entities 4000000 --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_4000000.json
$B --schema allhot --entities 2000000 --no-cpu-baseline 2>> $OUT/bench.err | J > $OUT/bench_allhot_2000000.json
$B --entities 16000000 --steps 30 --warmup 20 --no-cpu-baseline --no-extra --no-traffic 2>> $OUT/bench.err | J > $OUT/bench_16000000.json
for f in 3000000 4000000 allhot_2000000 16000000; do python - $OUT/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("parity"), d.get("roofline",{}).get("frac"))
PY
done
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-extra --no-traffic > $OUT/prof_stats.log 2>&1
find $OUT/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -3 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof_stats -type f ! -name '*stats.csv' -delete 2>/dev/null
