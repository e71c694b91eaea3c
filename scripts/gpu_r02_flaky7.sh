#!/bin/bash
# does the content VRAM holds before the process matter?  zero / 0xA5-fill most of VRAM in a separate process, then run the prefix
set -u
OUT=gpurun_out/r02fc; mkdir -p $OUT; export TMPDIR=/tmp
fill() { python - $1 <<'PY'
import sys, torch
v = int(sys.argv[1])
free, total = torch.cuda.mem_get_info()
n = int(free * 0.92)
x = torch.full((n,), v, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
print("filled", n >> 20, "MiB with", v)
PY
}
run() { name=$1; timeout 900 python -m pytest tests/test_box_game.py tests/test_cpp_host.py tests/test_despawn_rollback.py tests/test_gpu_custom_system.py tests/test_gpu_gen_groups.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/f7_$name.txt 2>&1; echo "$name: $(grep -E 'passed|failed' $OUT/f7_$name.txt | tail -n 1) $(grep -h 'AssertionError: frame' $OUT/f7_$name.txt | head -n 1)"; }
echo "boot $(cat /proc/sys/kernel/random/boot_id | cut -c1-8)"
run asis
fill 0; run after_zero
fill 165; run after_a5
fill 0; run after_zero2
