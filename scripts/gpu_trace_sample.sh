#!/bin/bash
# Sample of the request tracing: GGRS_HIP_TRACE lines for two SyncTest ticks, and roctx ranges in a rocprofv3 marker trace.
OUT=gpurun_out/${1:-trace}
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/trace_demo.py <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bevy_ggrs_amd as bg, common as cm
w = bg.World(5000, max_depth=8)
ids = cm.build_particles(w)
vel, ttl = cm.synthetic_particles(3000, ttl="despawn")
cm.spawn_particles(w, ids, 3000, vel, ttl)
drv = cm.SyncTestDriver(w, 2)
for _ in range(5):
    drv.tick((0,))
PY
GGRS_HIP_TRACE=1 timeout 100 python /tmp/trace_demo.py 2> $OUT/trace_lines.txt; tail -12 $OUT/trace_lines.txt
GGRS_HIP_ROCTX=1 timeout 200 rocprofv3 --marker-trace --kernel-trace -f csv -d $OUT/prof_marker -o m -- python /tmp/trace_demo.py > $OUT/prof_marker.log 2>&1
ls $OUT/prof_marker; head -12 $OUT/prof_marker/*marker_api_trace.csv 2>/dev/null | cut -c1-200
