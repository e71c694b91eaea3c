#!/usr/bin/env python
"""How many workgroups of a device-spawn world's kernel are REALLY resident?  Worlds of growing capacity, a few SyncTest ticks each; prints what the seal decided
and whether the rendezvous inside the launch completed.  usage: device_spawn_probe.py [capacity ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bevy_ggrs_amd as bg, common as cm
import test_gpu_device_spawn as t

for cap in [int(a) for a in sys.argv[1:]] or [360_000, 393_216, 400_000, 420_000, 440_000, 458_752]:
    out = {"capacity": cap}
    try:
        w = bg.World(cap, max_depth=6)
        t.build(w, cap // 4)
        out["device_spawn"] = w.kernel_info().get("device_spawn")
        drv = cm.SyncTestDriver(w, 4, max_prediction=5)
        t0 = time.perf_counter()
        for _ in range(6): drv.tick((0,))
        w.synchronize()
        out["ok"] = True; out["len"] = w.len; out["ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    except Exception as e:                                    # noqa: BLE001
        out["ok"] = False; out["error"] = str(e)[:200]
    print(json.dumps(out), flush=True)
