#!/bin/bash
# the driver's own round-end sequence on a fresh box: pytest -m gpu -x, smoke, default bench
set -u
TAG=${1:-a}
OUT=gpurun_out/r02soak; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_$TAG.txt 2>&1; echo "soak $TAG pytest rc=$? $(grep -E 'passed|failed' $OUT/pytest_$TAG.txt | tail -n 1)"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke_$TAG.txt 2>&1; echo "soak $TAG smoke rc=$?"
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "soak $TAG bench rc=$? $(python -c "import json; j=json.load(open('$OUT/bench_$TAG.json')); print('%.2f G' % (j['value']/1e9), 'frac %.3f' % j['roofline']['frac'], j['parity']['equal'])")"
