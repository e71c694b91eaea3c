#!/bin/bash
# r03zi: one specialised kernel per group shape (P2P sessions) + the interpreter's collector out of bench.py's timed regions.
# ONE gpurun call: the GPU suite, per-tick host times of a P2P-shaped session with / without specialisation (scripts/p2p_diag.py),
# BASELINE config 4 (waiting for the kernels / not / specialisation off), config 2, 100 k, the headline in both forms.
OUT=gpurun_out/r03zi; mkdir -p $OUT
timeout 400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_full.log
{
GGRS_JIT_SPECIALISE_AFTER=4 GGRS_JIT_SPECIALISE_SYNC=1 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=0 timeout 100 python scripts/p2p_diag.py
} > $OUT/p2p_diag.txt 2>&1
{
GGRS_JIT_SPECIALISE_AFTER=4 DIAG_SETTLE=1 timeout 100 python scripts/p2p_diag.py
GGRS_JIT_SPECIALISE_AFTER=4 DIAG_SETTLE=1 DIAG_TORCH=1 timeout 100 python scripts/p2p_diag.py
} > $OUT/p2p_diag_async.txt 2>&1
timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc.json
timeout 200 python bench.py --config 4 --no-cpu-baseline --no-specialise-wait 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc_nowait.json
GGRS_JIT_SPECIALISE_AFTER=0 timeout 200 python bench.py --config 4 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config4_gc_generic.json
timeout 200 python bench.py --config 2 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_config2.json
timeout 200 python bench.py --entities 100000 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_100000.json
timeout 300 python bench.py 2>> $OUT/bench.err | grep '^{' > $OUT/bench.json
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | grep '^{' > $OUT/bench_driver_form_$i.json; done
grep -E "passed|failed|rc=" $OUT/pytest_gpu_full.log | tail -3
