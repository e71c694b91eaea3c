#!/usr/bin/env python
"""Is k_tick's two-mode latency (~121 vs ~129 us at 1 M) a property of the arena's placement or of the process?
Creates several worlds in ONE process (holding the previous arenas so every world lands somewhere else) and reports
the steady-state k_tick time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, bevy_ggrs_amd as bg, common as cm
keep = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    w, ids = bench.build_world(bg, cm, 1_000_000, 8)
    bench.warm_ring(bg, w, 8)
    run, _k = bench.tick_requests(bg, w, 8)
    for _ in range(20): run(w.frame)
    w.profile_enable(True)
    for _ in range(60): run(w.frame)
    ms, n = w.profile_read()["tick"]
    ptr, ts = w.column_device_ptr(ids[0], 0)
    print("world %d: k_tick %.1f us  (column 0 at %#x)" % (k, ms / n * 1e3, ptr), flush=True)
    keep.append((w, run, _k))          # keep the arena allocated: the next world gets a different placement
