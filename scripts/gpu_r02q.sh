#!/bin/bash
OUT=gpurun_out/${1:-r02q}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v 'RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' | tail -4 > $OUT/pytest.log; cat $OUT/pytest.log
for n in 10000 100000 300000; do ./benches/tick_bench $n 8 300 16 0 0 1; ./benches/tick_bench $n 8 300 16 0 1 1; done 2>&1 | tee $OUT/tb.txt
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof10k -o t -- ./benches/tick_bench 10000 8 300 16 0 0 1 > $OUT/prof10k.log 2>&1
python - <<'PY'
import csv,glob,statistics
f=glob.glob("gpurun_out/r02q/prof10k/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))))
ticks=[r for r in rows if "k_tick1" in r[2]]
d=[e-s for s,e,_ in ticks][50:250]
gaps=[ticks[i+1][0]-ticks[i][1] for i in range(50,250)]
print("k_tick1 dur us mean %.2f min %.2f  gap between consecutive k_tick1 mean %.2f min %.2f"%(statistics.mean(d)/1e3,min(d)/1e3,statistics.mean(gaps)/1e3,min(gaps)/1e3))
PY
