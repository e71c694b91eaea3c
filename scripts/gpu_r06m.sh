#!/bin/bash
# round 6, call m: device-spawn rendezvous through mailbox words (was: a counter barrier with an acquire in the poll loop); residency bounded by the code object's
# own VGPR / SGPR counts; the C-ABI branch-step test
out=gpurun_out/r06m; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_device_spawn.py tests/test_gpu_zfanout.py::test_branch_step_with_a_user_written_spawner_through_the_c_abi tests/test_gpu_jit_cache.py tests/test_gpu_custom_system.py -q -m gpu 2>&1 | tail -60 > $out/pytest.log; echo "pytest rc=$?"; tail -4 $out/pytest.log | cut -c1-400
timeout 300 python scripts/device_spawn_probe.py 360000 393216 400000 > $out/probe.txt 2>&1
timeout 600 python scripts/device_spawn_bench.py 70000 95000 > $out/device_spawn_session.txt 2>&1; cat $out/device_spawn_session.txt | cut -c1-600
timeout 300 python bench.py --no-traffic > $out/bench.json 2> $out/bench.err; cut -c1-260 $out/bench.json
