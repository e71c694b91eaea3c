#!/bin/bash
# round 6: value-tag ids that start over (every tag-keeping world now crosses its first start-over within a few dozen launches): the new test, the value-tag files and fuzzes
out=gpurun_out/r06kk; mkdir -p $out
T="python -m pytest -q -m gpu -p no:cacheprovider"
timeout 500 $T tests/test_gpu_row_versions.py tests/test_gpu_zfuzz_branches.py tests/test_gpu_device_spawn.py tests/test_fuzz_requests.py -k "value_tag or row_version or branch or spawn" 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -6 | cut -c1-300 | tee $out/tags_wrap.log
timeout 200 python bench.py --entities 4000000 --no-cpu-baseline --no-extra --no-traffic 2> $out/bench.err | grep '^{' > $out/bench_4000000.json; python - $out/bench_4000000.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print("4 M:", d["value"], d["ms_per_step"], d.get("parity"))
PY
