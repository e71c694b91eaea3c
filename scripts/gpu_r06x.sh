#!/bin/bash
# round 6, call x: self-fold for blocking calls on the final form -- the whole suite, smoke, the driver's bench command
out=gpurun_out/${1:-r06x}; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu --durations=5 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -30 > $out/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -12 $out/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench.err; echo "bench rc=$? stdout lines: $(wc -l < $out/bench_driver_form.json)"; cut -c1-260 $out/bench_driver_form.json
