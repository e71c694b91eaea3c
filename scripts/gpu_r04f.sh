#!/bin/bash
# Round 4, call F: hunt for the replica desync seen once in `bench.py --gpus 2 --oversubscribe --entities 300000` (r04e).
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt
DBL=$PWD/tests/cpp/_build/librccl_double.so
run2() {  # name, env, extra args
  local ok=0 bad=0
  for i in 1 2 3 4 5 6; do
    if env $2 GGRS_RCCL_LIB=$DBL timeout 300 python bench.py --gpus 2 --oversubscribe --steps 6 --warmup 2 --preheat-ms 20 --entities 300000 --no-cpu-baseline $3 > $OUT/run_$1_$i.out 2> $OUT/run_$1_$i.err; then ok=$((ok+1)); else bad=$((bad+1)); grep -h "Desync\|Error" $OUT/run_$1_$i.err | tail -2; fi
  done
  echo "== $1: ok=$ok bad=$bad" | tee -a $OUT/summary.txt
}
run2 default "A=1" ""
run2 nospec "GGRS_JIT_SPECIALISE_AFTER=0" ""
run2 nolanefold "GGRS_JIT_LANE_FOLD=0" ""
run2 hostfoldoff_groupfold "GGRS_GROUP_FOLD_MIN_WGS=8" ""
run2 nt "A=1" "--nt"
# one rank, every step against the oracle (no transport involved): nondeterminism inside one process would show here
for i in 1 2 3; do
  timeout 600 python bench.py --fanout --entities 300000 --steps 300 --warmup 0 --preheat-ms 0 --parity-steps 300 --cpu-ticks 1 2> $OUT/ws1_$i.err | grep '^{' > $OUT/ws1_$i.json; echo "ws1 run $i rc=$? $(python -c "import json;j=json.load(open('$OUT/ws1_$i.json'));print(j['parity'])" 2>&1 | cut -c1-300)" | tee -a $OUT/summary.txt
done
# the same single rank while a second process loads the GPU
( timeout 200 python bench.py --schema allhot --steps 400000 --no-cpu-baseline --preheat-ms 0 > /dev/null 2>&1 & )
sleep 12
for i in 1 2; do
  timeout 600 python bench.py --fanout --entities 300000 --steps 300 --warmup 0 --preheat-ms 0 --parity-steps 300 --cpu-ticks 1 2> $OUT/ws1_load_$i.err | grep '^{' > $OUT/ws1_load_$i.json; echo "ws1 under load run $i rc=$? $(python -c "import json;j=json.load(open('$OUT/ws1_load_$i.json'));print(j['parity'])" 2>&1 | cut -c1-300)" | tee -a $OUT/summary.txt
done
wait
cat $OUT/summary.txt
