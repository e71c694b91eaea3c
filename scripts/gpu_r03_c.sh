#!/bin/bash
# round 3, call C: full GPU suite on the refactored library + new schema tests, ALU microbenchmark, headline A/B of the two kernel families
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
./scripts/ubench_alu > $O/ubench_alu.txt 2>&1; cat $O/ubench_alu.txt
B="python bench.py --no-cpu-baseline --steps 100 --warmup 16 --arena paged"
$B > $O/bench_tick3.json 2> $O/bench_tick3.err
GGRS_TICK_GENERIC=1 $B > $O/bench_jit_persist.json 2> $O/bench_jit_persist.err
GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0 $B > $O/bench_jit_tiles.json 2> $O/bench_jit_tiles.err
GGRS_ROW_VERSIONS=0 GGRS_TICK_GENERIC=1 $B > $O/bench_jit_persist_fullcopy.json 2> $O/bench_jit_persist_fullcopy.err
GGRS_ROW_VERSIONS=0 GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0 $B > $O/bench_jit_tiles_fullcopy.json 2> $O/bench_jit_tiles_fullcopy.err
$B --entities 4000000 > $O/bench_tick3_4m.json 2> $O/bench_tick3_4m.err
GGRS_TICK_GENERIC=1 $B --entities 4000000 > $O/bench_jit_persist_4m.json 2> $O/bench_jit_persist_4m.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    r=j["roofline"]; print(" value %.2fG ms/step %.4f frac %.3f avg_launch %.1f B/ent %.0f kernel %s" % (j["value"]/1e9, j["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("algorithmic_bytes_per_entity",0), j["config"].get("request_group_kernel")))
    print("  other", r.get("other_kernels"), " parity", j.get("parity"))
except Exception as e: print("  ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
