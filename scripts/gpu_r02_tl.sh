#!/bin/bash
# timeline of a pipelined 10 k / 50 k / 100 k tick on the final code: kernel durations and gaps (rocprofv3 kernel trace)
set -u
OUT=gpurun_out/r02hf; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 10000 50000 100000; do
  rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_$n -- $R/benches/tick_bench $n 8 400 50 0 0 1 > $R/$OUT/trace_$n.log 2>&1
  python - $R/$OUT/trace_$n $n <<'PY' | tee -a $R/$OUT/timeline.txt
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
print('entities', sys.argv[2], '(under rocprofv3: the host side is slower than in a plain run)')
for k, v in dur.items(): print('  dur ', k, len(v), 'avg %.2f us' % (sum(v) / len(v) / 1e3), 'min %.2f' % (min(v) / 1e3))
for k, v in gaps.items(): print('  gap ', k, len(v), 'avg %.2f us' % (sum(v) / len(v) / 1e3), 'min %.2f' % (min(v) / 1e3))
PY
  rm -rf $R/$OUT/trace_$n
done
