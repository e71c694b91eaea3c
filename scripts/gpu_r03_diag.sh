#!/bin/bash
# round 3, call A: the driver's exact bench command next to the 200/16 form, with and without the pre-heat; telemetry in the JSON line
O=gpurun_out/r03a; mkdir -p $O
rocm-smi --showclocks --showpower > $O/smi_start.txt 2>&1
ls -la /sys/class/drm/ > $O/sysfs.txt 2>&1; cat /sys/class/drm/card*/device/pp_dpm_sclk >> $O/sysfs.txt 2>&1
python -c "import torch; print(torch.cuda.device_count())" > $O/ndev.txt 2>&1
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $O/drv_nopre_$i.json 2> $O/drv_nopre_$i.err; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_pre_$i.json 2> $O/drv_pre_$i.err; done
python bench.py --gpus 1 --steps 200 --warmup 16 --preheat-ms 0 --no-cpu-baseline > $O/b200_nopre.json 2> $O/b200_nopre.err
python bench.py --gpus 1 --steps 200 --warmup 16 --no-cpu-baseline > $O/b200_pre.json 2> $O/b200_pre.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_full.json 2> $O/drv_full.err
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    r=j["roofline"]; print(" value %.2fG ms/step %.4f frac %.3f avg_launch %.1f arena %s preheat %s" % (j["value"]/1e9, j["ms_per_step"], r["frac"], r["avg_launch_us"], j["config"].get("arena_actual"), j.get("preheat")))
    print("  launch_us", r.get("launch_us")); print("  tick_wall", j.get("telemetry",{}).get("tick_wall_us")); print("  clocks", j.get("telemetry",{}).get("clocks_start"), j.get("telemetry",{}).get("clocks_end"))
    print("  contig", r.get("contig_arena_variant")); print("  parity", j.get("parity"))
except Exception as e: print("  ERR", e)
PY
done
