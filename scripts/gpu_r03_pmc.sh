#!/bin/bash
# counter passes (9 rocprofv3 --pmc runs each, scripts/pmc_passes.py) of the generated kernel's two forms on the headline world (320 B/entity-tick);
# round 3 ran a third arm, k_tick3, which no longer exists (profiles/r03p)
set -u
TAG=${1:-r03p}; OUT=gpurun_out/$TAG; mkdir -p $OUT/jit_tiles $OUT/jit_persist; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_schema.py -x -q -m gpu > $OUT/pytest_schema.log 2>&1; tail -3 $OUT/pytest_schema.log
for v in jit_tiles jit_persist; do
  case $v in
    jit_tiles) ENVV="GGRS_JIT_PERSIST_MIN_SLOTS=0";;
    jit_persist) ENVV="GGRS_JIT_PERSIST_MIN_SLOTS=1";;
  esac
  echo "== $v: $(env $ENVV timeout 120 benches/tick_bench 1000000 8 100 16 0 0 1 2>&1 | tail -n 1 | cut -c1-300)" | tee -a $OUT/plain.txt
  env $ENVV PMC_MAX_PASSES=9 timeout 900 python scripts/pmc_passes.py $OUT/$v $OUT/counters.txt -- ./benches/tick_bench 1000000 8 40 8 0 1 1 > $OUT/${v}_passes.log 2>&1
  rm -rf $OUT/$v/pmc_*
done
python - <<'PY'
import json
for v in ("jit_tiles","jit_persist"):
    try:
        j=json.load(open(f"gpurun_out/%s/%s/pmc_counters.json" % (__import__("os").environ.get("TAG","r03p"), v)))
    except Exception as e:
        print(v, "ERR", e); continue
    for kn,d in j["kernels"].items():
        if "tick" not in kn: continue
        keys=["dispatches","GRBM_GUI_ACTIVE","SQ_WAVES","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_INSTS_VALU","SQ_INSTS_VMEM_WR","SQ_INST_CYCLES_VMEM_WR","TCC_EA0_WRREQ_sum","TCC_EA0_WRREQ_64B_sum","TCC_EA0_WRREQ_STALL_sum","TCC_EA0_RDREQ_sum","TCC_BUSY_sum","TCC_CYCLE_sum","TCC_TAG_STALL_sum","TCC_WRITEBACK_sum","TCP_PENDING_STALL_CYCLES_sum","TA_TA_BUSY_sum"]
        print(v, kn[:40], {k: round(d[k]) for k in keys if k in d})
PY
