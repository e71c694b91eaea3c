#!/usr/bin/env python
"""A SyncTest session that holds INPUT_SPAWN (the stress_test with the spawn key down: 100 particles per frame, particles.rs:258-270), depth 8,
100 k entities to start with: per-tick time with the spawn fused into the request group (default) and on the one-launch-per-request path, GGRS_TICK_JIT=0 (a firing
spawn system ends the group: every resimulated frame is its own launches).  Every checksum is compared with the CPU oracle."""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n=100_000, D=8, ticks=120):
    import numpy as np
    import bevy_ggrs_amd as bg, common as cm
    from oracle.binding import FLAT, OracleWorld
    out = {}
    for name, w in (("gpu", bg.World(n + 100 * (ticks + 2 * D + 4), max_depth=D + 1)), ("oracle", OracleWorld(n + 100 * (ticks + 2 * D + 4), D + 1, FLAT))):
        ids = cm.build_particles(w, with_spawn=True)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
        fn = cm.frame_spawn_fn(100)
        for _ in range(D + 2): drv.tick((cm.INPUT_SPAWN,), spawn_fn=fn)
        if name == "gpu": w.synchronize()
        t0 = time.perf_counter()
        for _ in range(ticks): drv.tick((cm.INPUT_SPAWN,), spawn_fn=fn)
        if name == "gpu": w.synchronize()
        out[name] = (time.perf_counter() - t0, drv.all_checksums, w.kernel_info().get("spawn_system") if name == "gpu" else None, w.len)
    assert out["gpu"][1] == out["oracle"][1], "checksums differ"
    return {"us_per_tick": out["gpu"][0] / ticks * 1e6, "oracle_us_per_tick": out["oracle"][0] / ticks * 1e6, "spawn_system": out["gpu"][2], "len_end": out["gpu"][3],
            "saves_checked": len(out["gpu"][1]), "ticks": ticks, "entities_start": n, "depth": D}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(json.dumps(run()))
    else:
        for env in ({}, {"GGRS_TICK_JIT": "0"}):                     # the spawn inside the request group's launch / one launch per request
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env={**os.environ, **env})
            print(env or "default", r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-2000:])
