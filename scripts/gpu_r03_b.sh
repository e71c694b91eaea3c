#!/bin/bash
# round 3, call B: the refactored library -- full GPU suite, then the headline bench A/B (row versions on/off, k_tick3 vs generated persistent)
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --steps 100 --warmup 16"
$B > $O/bench_default.json 2> $O/bench_default.err
GGRS_ROW_VERSIONS=0 $B --arena paged > $O/bench_fullcopy.json 2> $O/bench_fullcopy.err
GGRS_TICK_GENERIC=1 $B --arena paged > $O/bench_jit.json 2> $O/bench_jit.err
GGRS_TICK_GENERIC=1 GGRS_ROW_VERSIONS=0 $B --arena paged > $O/bench_jit_fullcopy.json 2> $O/bench_jit_fullcopy.err
GGRS_TICK_GENERIC=1 GGRS_JIT_PERSIST_MIN_SLOTS=0 GGRS_ROW_VERSIONS=0 $B --arena paged > $O/bench_jit_tiles_fullcopy.json 2> $O/bench_jit_tiles_fullcopy.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    r=j["roofline"]; print(" value %.2fG ms/step %.4f frac %.3f avg_launch %.1f B/ent %.0f arena %s kernel %s" % (j["value"]/1e9, j["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("algorithmic_bytes_per_entity",0), j["config"].get("arena_actual"), j["config"].get("request_group_kernel")))
    print("  other", r.get("other_kernels")); print("  contig", r.get("contig_arena_variant")); print("  parity", j.get("parity"))
except Exception as e: print("  ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
