#!/bin/bash
# final DP policy: full GPU suite + sizes sweep, and a timeline of the generic kernel at 100k
set -u
OUT=gpurun_out/r02dq; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
for n in 10000 30000 50000 70000 100000 150000 200000 300000; do
  echo "n=$n $(timeout 120 benches/tick_bench $n 8 400 50 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $OUT/sizes.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -n 4 $OUT/pytest.txt
cd /tmp; export TMPDIR=/tmp
for n in 100000; do
  GGRS_TICK_GENERIC=1 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_$n -- $R/benches/tick_bench $n 8 400 50 0 0 1 > $R/$OUT/trace_$n.log 2>&1
  python - $R/$OUT/trace_$n <<'PY' | tee -a $R/$OUT/timeline_gen.txt
import sys, glob, csv
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//3:]
dur = {}; gaps = {}; prev = None
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur.setdefault(k, []).append(e - s)
    if prev is not None: gaps.setdefault(prev[0] + ' -> ' + k, []).append(s - prev[1])
    prev = (k, e)
print(sys.argv[1])
for k, v in dur.items(): print('  dur ', k, len(v), 'avg %.2f us' % (sum(v) / len(v) / 1e3), 'min %.2f' % (min(v) / 1e3))
for k, v in gaps.items(): print('  gap ', k, len(v), 'avg %.2f us' % (sum(v) / len(v) / 1e3), 'min %.2f' % (min(v) / 1e3))
PY
  rm -rf $R/$OUT/trace_$n
  echo "gen n=$n $(GGRS_TICK_GENERIC=1 timeout 120 $R/benches/tick_bench $n 8 400 50 0 0 1 2>&1 | tail -n 1 | cut -c1-260)" | tee -a $R/$OUT/sizes.txt
done
