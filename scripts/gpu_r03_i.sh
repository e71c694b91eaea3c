#!/bin/bash
O=gpurun_out/${1:-r03i}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_row_versions.py tests/test_gpu_zfanout.py tests/test_gpu_knobs.py -x -q -m gpu > $O/pytest_some.log 2>&1
tail -8 $O/pytest_some.log
python bench.py --no-checksum --no-cpu-baseline > $O/bench_nochecksum.json 2> $O/err.txt
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03i/bench_nochecksum.json")); r=j["roofline"]
print("no-checksum: ms/step %.4f kernel %.1f us B/ent %s" % (j["ms_per_step"], r["avg_launch_us"], r.get("algorithmic_bytes_per_entity")))
PY
