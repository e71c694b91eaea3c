#!/usr/bin/env python
"""What a tick costs a world whose systems SPAWN ON THE DEVICE (tests/test_gpu_device_spawn.py's splitting cells, e.spawn(n) + GGRS_SPAWN_PAYLOAD_PARENT): SyncTest
depth 8, one cooperative launch per tick with a grid barrier per simulated frame (two in frames that spawn).  Timing only -- parity with the oracle is the test's job.
Cells are re-seeded with long fuses so that the population splits throughout the run.  usage: device_spawn_bench.py [entities ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bevy_ggrs_amd as bg, common as cm
import test_gpu_device_spawn as t


def run(n, D=8, ticks=150):
    w = bg.World(4 * n + 256, max_depth=D + 1)
    cell = w.register_component("Cell", 4, 4)
    w.checksum_component(cell, [0, 1, 2, 3])
    w.add_custom_system(t.SPLIT_SRC, [(cell, 0), (cell, 1), (cell, 2), (cell, 3)], iparam=(1,), name="split")
    w.add_spawn_system(t.CHILD_SRC, [cell], [(cell, 0), (cell, 1), (cell, 2), (cell, 3)], payload_stride=t.PARENT, name="child")
    rng = np.random.default_rng(7)
    w.spawn(n, {cell: [rng.uniform(-50, 50, n).astype(np.float32).view(np.uint32), rng.uniform(-9, 9, n).astype(np.float32).view(np.uint32),
                       (20 + np.arange(n, dtype=np.uint32) % (ticks + 40)).astype(np.uint32), np.zeros(n, dtype=np.uint32)]})
    drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
    for _ in range(D + 8): drv.tick((0,))
    w.synchronize(); len0 = w.len
    t0 = time.perf_counter()
    for _ in range(ticks): drv.tick((0,))
    w.synchronize()
    secs = time.perf_counter() - t0
    w.profile_enable(True)
    for _ in range(20): drv.tick((0,))
    prof = w.profile_read(); w.profile_enable(False)
    return {"entities_start": n, "capacity": 4 * n + 256, "depth": D, "ticks": ticks, "us_per_tick_blocking_api": round(secs / ticks * 1e6, 1), "len_start": len0, "len_end": w.len,
            "kernel_us_mean": round(prof["tick"][0] / max(prof["tick"][1], 1) * 1e3, 2), "launches_per_tick": prof["tick"][1] / 20,
            "entity_frames_per_s": round((len0 + w.len) / 2 * (D + 1) * ticks / secs / 1e9, 2), "unit": "G entity-frames/s (mean len x 9 AdvanceWorlds per tick)"}


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [70_000, 120_000]:
        print(json.dumps(run(n)))
