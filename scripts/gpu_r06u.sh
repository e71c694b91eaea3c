#!/bin/bash
# round 6, call u: the whole GPU suite, smoke and the driver's bench command on the final tree (after the fan-out's refusal paths and the device-spawn epoch start-over)
out=gpurun_out/r06u; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu --durations=5 -p no:cacheprovider 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -25 > $out/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench.err; echo "bench rc=$? stdout lines: $(wc -l < $out/bench_driver_form.json)"; cut -c1-330 $out/bench_driver_form.json
