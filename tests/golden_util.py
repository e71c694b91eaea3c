"""Shared by tests/test_golden.py (oracle, CPU) and tests/test_gpu_golden.py (HIP, GPU): replays
the committed known-answer cases of tests/golden/hot_path_vectors.json on any backend world."""
import json
import os

import numpy as np

import common as cm

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "hot_path_vectors.json")))


def fold(a):
    h = 0xcbf29ce484222325
    for v in np.asarray(a, dtype=np.uint64).tolist():
        h = ((h ^ v) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def replay_particles_case(make_world, case):
    a = case["args"]
    n, cd, ticks = a["n"], a["check_distance"], a["ticks"]
    w = make_world(n + a["rate"] * ticks + 64, 16)
    ids = cm.build_particles(w, with_spawn=True, ttl_init=a["ttl_init"])
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, cd)
    fn = cm.frame_spawn_fn(a["rate"])
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if t % a["spawn_every"] == 1 else 0,), spawn_fn=fn)
    got = [[int(f), f"{c:032x}"] for f, c in drv.all_checksums]
    assert got == case["checksums"]
    st = cm.snapshot_state(w, ids)
    fin = case["final"]
    assert (int(st["len"]), int(st["frame"]), int(st["alive"].sum())) == (fin["len"], fin["frame"], fin["active"])
    for k, want in fin["folds"].items():
        assert f"{fold(st[k]):016x}" == want, k
