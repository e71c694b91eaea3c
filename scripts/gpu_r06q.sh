#!/bin/bash
# round 6, call q: the whole GPU suite + smoke on the final tree; the device-spawn tests three more times (the rendezvous are bounded waits: look for flakes)
out=gpurun_out/r06q; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -25 > $out/pytest_gpu.log; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_device_spawn.py -q -m gpu 2>&1 | tail -1; done | tee $out/devspawn_x3.log
