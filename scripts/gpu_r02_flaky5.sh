#!/bin/bash
set -u
OUT=gpurun_out/r02fb; mkdir -p $OUT; export TMPDIR=/tmp
rep() { name=$1; shift; "$@" > $OUT/f5_$name.txt 2>&1; echo "$name: $(grep -E 'passed|failed' $OUT/f5_$name.txt | tail -n 1) $(grep FAILED $OUT/f5_$name.txt | tr '\n' ' ')" | tee -a $OUT/f5.txt; }
rm -rf tests/cpp/_build
rep rebuild1 timeout 900 python -m pytest tests -m gpu -q
rm -rf tests/cpp/_build
rep rebuild2 timeout 900 python -m pytest tests -m gpu -q
rm -rf tests/cpp/_build
rep rebuild3_nocpp_before timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_cpp_host.py
rep sleep_then_parity bash -c 'python - <<PY
import time, subprocess, sys
time.sleep(50)
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "tests/test_gpu_golden.py", "tests/test_gpu_parity.py", "-m", "gpu", "-q"]))
PY'
