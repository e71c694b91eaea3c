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


def _state_folds(st):
    return {k: f"{fold(np.asarray(v).astype(np.uint64)):016x}" for k, v in sorted(st.items())
            if isinstance(v, np.ndarray)}


# ---- scenario runners shared by the generator (oracle) and the replays (oracle on CPU, HIP on GPU)
def run_box_game(make_world, a):
    from test_box_game import synctest_box_game
    cs, trace = synctest_box_game(make_world(a["n"] + 8, 8), a["players"], a["check_distance"], a["ticks"], a["n"])
    t, v = trace[-1]
    return {"n_checksums": len(cs), "checksums_head": [[int(f), f"{c:032x}"] for f, c in cs[:12]],
            "translation_fold": f"{fold(t.reshape(-1)):016x}", "velocity_fold": f"{fold(v.reshape(-1)):016x}",
            "cube0": [f"{int(x):08x}" for x in list(t[0]) + list(v[0])]}


def run_despawn_rollback(make_world, a):
    from test_despawn_rollback import synctest_run
    cs, trace = synctest_run(make_world(a["n"] + 56, 8), a["n"], a["ticks"], a["check_distance"])
    mid, last = trace[a["ticks"] // 3], trace[-1]
    return {"checksums": [[int(f), f"{c:032x}"] for f, c in cs], "mid": _state_folds(mid), "last": _state_folds(last)}


def run_p2p_shape(make_world, a):
    from test_oracle_selfcheck import _p2p_run
    drv, st = _p2p_run(make_world(a["n"] + 40 * (a["ticks"] + 10) + 64, 8), a["n"], a["ticks"])
    return {"depths": [int(d) for d in drv.depths], "checksums": [[int(f), f"{c:032x}"] for f, c in drv.all_checksums],
            "final": _state_folds(st), "len": int(st["len"]), "frame": int(st["frame"])}


def run_allhot_spawn_held(make_world, a):
    """The all-columns-hot schema (tests/common.py: stress_test systems + increase_component over rotation / scale) under a SyncTest session that
    HOLDS the spawn key: every frame, resimulated ones included, spawns `rate` particles -- on the HIP path inside the request group's launch."""
    n, cd, ticks, rate = a["n"], a["check_distance"], a["ticks"], a["rate"]
    w = make_world(n + rate * (ticks + 2 * cd + 4), cd + 1)
    ids = cm.build_particles(w, with_spawn=True, ttl_init=a["ttl_init"], schema="allhot")
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, cd, max_prediction=cd + 1)
    fn = cm.frame_spawn_fn(rate)
    for _ in range(ticks):
        drv.tick((cm.INPUT_SPAWN,), spawn_fn=fn)
    st = cm.snapshot_state(w, ids)
    return {"checksums": [[int(f), f"{c:032x}"] for f, c in drv.all_checksums], "final": _state_folds(st), "len": int(st["len"]), "frame": int(st["frame"]),
            "active": int(st["alive"].sum())}


SCENARIOS = {"box_game_synctest": run_box_game, "despawn_rollback_synctest": run_despawn_rollback, "p2p_shape": run_p2p_shape,
             "allhot_spawn_held": run_allhot_spawn_held}


def replay_scenario(make_world, kind, case):
    got = SCENARIOS[kind](make_world, case["args"])
    assert got == case["expect"], {k: (got[k], case["expect"][k]) for k in got if got[k] != case["expect"][k]}
