#!/usr/bin/env python
"""Generates tests/golden/*.json -- frozen known-answer vectors for the hot path.

The reference (Rust) cannot be built or imported in this image and its own tests hold no absolute
checksum / f32 values, so these vectors are produced by the CPU oracle
(oracle/ggrs_oracle.cpp) and cross-checked against its independent numpy twin
(oracle/oracle_np.py) at generation time; the only externally published anchor is seahash's
documented vector.  Once committed they pin BOTH the oracle (tests/test_golden.py, CPU) and the
HIP path (tests/test_gpu_golden.py, GPU, no oracle involved at run time) against regressions.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common as cm  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
from oracle.binding import FLAT, REFSHAPED, OracleWorld, lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def fold(a):
    """order-sensitive 64-bit fold of a word column (FNV-style), cheap to recompute anywhere"""
    h = 0xcbf29ce484222325
    for v in np.asarray(a, dtype=np.uint64).tolist():
        h = ((h ^ v) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def particles_case(n, cd, ticks, spawn_every, rate, ttl_init, mode):
    w = OracleWorld(n + rate * ticks + 64, 16, mode)
    ids = cm.build_particles(w, with_spawn=True, ttl_init=ttl_init)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, cd)
    fn = cm.frame_spawn_fn(rate)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if t % spawn_every == 1 else 0,), spawn_fn=fn)
    st = cm.snapshot_state(w, ids)
    return drv.all_checksums, st


def main():
    out = {}
    # ---- seahash: the crate's published vector + derived vectors of the formulas on this path
    msg = b"to be or not to be"
    import ctypes
    buf = (ctypes.c_uint8 * len(msg))(*msg)
    out["seahash"] = {
        "published_to_be_or_not_to_be": int(lib.gor_seahash_buffer(buf, len(msg))),
        "entity_checksum_active1_total1": int(lib.gor_entity_checksum(1, 1)),
        "finalize_part_0": int(lib.gor_finalize_part(0)),
        "entity_part_order7_inner9": int(lib.gor_entity_part(7, 9)),
        "diffuse_1": int(lib.gor_diffuse(1)),
    }
    assert out["seahash"]["published_to_be_or_not_to_be"] == 1988685042348123509
    assert out["seahash"]["published_to_be_or_not_to_be"] == onp.seahash_buffer(msg)
    # ---- Time<GgrsTime> delta bits (src/time.rs:63-87) for frames 1..12 at 60 and 50 fps
    out["dt_bits"] = {str(fps): [int(lib.gor_dt_bits(fps, f)) for f in range(1, 13)] for fps in (60, 50, 144)}
    for fps, v in out["dt_bits"].items():
        assert v == [onp.dt_bits(int(fps), f) for f in range(1, 13)], "C++ oracle and numpy twin disagree on dt"
    assert out["seahash"]["entity_checksum_active1_total1"] == onp.entity_checksum(1, 1) == 0x7c846906b6e5a068
    assert out["seahash"]["diffuse_1"] == onp.diffuse(1)
    # ---- particles SyncTest runs (checksum of EVERY SaveGameState in order + folds of the final world)
    cases = {}
    for name, (n, cd, ticks, se, rate, ttl0) in {
        "n257_cd3": (257, 3, 12, 3, 100, 40),
        "n1_cd2": (1, 2, 9, 4, 3, 5),
        "n5000_cd7": (5000, 7, 20, 3, 100, 40),
    }.items():
        cs, st = particles_case(n, cd, ticks, se, rate, ttl0, FLAT)
        cs2, st2 = particles_case(n, cd, ticks, se, rate, ttl0, REFSHAPED)
        assert cs == cs2, "flat and reference-shaped oracle disagree"
        cm.assert_states_equal(st, st2, name)
        cases[name] = {
            "args": {"n": n, "check_distance": cd, "ticks": ticks, "spawn_every": se, "rate": rate, "ttl_init": ttl0},
            "checksums": [[int(f), f"{c:032x}"] for f, c in cs],
            "final": {"len": int(st["len"]), "frame": int(st["frame"]), "active": int(st["alive"].sum()),
                      "folds": {k: f"{fold(v):016x}" for k, v in st.items() if isinstance(v, np.ndarray) and k.startswith("c")}},
        }
    out["particles_synctest"] = cases
    # ---- further scenarios (SURVEY 8f rows + BASELINE configs 1 and 4): both oracle shapes must agree before freezing
    import golden_util as gu
    scen = {
        "box_game_synctest": {"2p_cd7": {"n": 2, "players": 2, "check_distance": 7, "ticks": 40},
                              "300cubes_4p_cd3": {"n": 300, "players": 4, "check_distance": 3, "ticks": 25}},
        "despawn_rollback_synctest": {"n200_cd3": {"n": 200, "ticks": 16, "check_distance": 3},
                                      "n5000_cd2": {"n": 5000, "ticks": 14, "check_distance": 2}},
        "p2p_shape": {"n600": {"n": 600, "ticks": 60}, "n3000": {"n": 3000, "ticks": 80}},
        "allhot_spawn_held": {"n300_cd3": {"n": 300, "check_distance": 3, "ticks": 14, "rate": 70, "ttl_init": 9},
                              "n9000_cd8": {"n": 9000, "check_distance": 8, "ticks": 18, "rate": 100, "ttl_init": 40}},
    }
    out["scenarios"] = {}
    for kind, cases_ in scen.items():
        out["scenarios"][kind] = {}
        for name, args in cases_.items():
            a = gu.SCENARIOS[kind](lambda cap, depth: OracleWorld(cap, depth, FLAT), args)
            b = gu.SCENARIOS[kind](lambda cap, depth: OracleWorld(cap, depth, REFSHAPED), args)
            assert a == b, (kind, name, "flat and reference-shaped oracle disagree")
            out["scenarios"][kind][name] = {"args": args, "expect": a}
    with open(os.path.join(HERE, "hot_path_vectors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "hot_path_vectors.json"))


if __name__ == "__main__":
    main()
