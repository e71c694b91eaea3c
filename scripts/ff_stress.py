#!/usr/bin/env python
"""Fold-forward publication stress: a long pipelined SyncTest session on a fold-forward world (the NEXT launch folds a tick's checksum
rows and publishes every value as ONE 16-byte {value, tag} store into pinned memory, which the collecting host polls at full speed) next to a SHADOW world that
runs the same ticks with the fold on the host (GGRS_FOLD_FORWARD_MIN_WGS=1000000: no tags, no polling of device-published cells): every Checksum(u128) of
every tick must be equal, and -- SyncTest -- every frame's checksum equal in every tick that re-simulates it.  A torn or reordered publication is a mismatch.
With --sync every tick is a BLOCKING ggrs_hip_handle_requests instead: the fold-forward world then folds its own rows inside the launch (self-fold: the tile
workgroups' 16-byte {value, tag} cells, read by the launch's own fold workgroups as they arrive), the shadow uses k_gen_finalize.
usage: ff_stress.py [entities] [ticks] [--sync]      prints one JSON line"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bevy_ggrs_amd as bg
import common as cm


def world(n, D, host_fold):
    if host_fold: os.environ["GGRS_FOLD_FORWARD_MIN_WGS"] = "1000000"
    else: os.environ.pop("GGRS_FOLD_FORWARD_MIN_WGS", None)
    w = bg.World(n, max_depth=D + 1)
    os.environ.pop("GGRS_FOLD_FORWARD_MIN_WGS", None)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(D + 1); w.set_synctest_check_distance(D)
    return w


def main():
    sync = "--sync" in sys.argv
    argv = [a for a in sys.argv if a != "--sync"]
    n = int(argv[1]) if len(argv) > 1 else 300_000
    ticks = int(argv[2]) if len(argv) > 2 else 1_000_000
    D = 8
    ws = [world(n, D, False), world(n, D, True)]
    assert "fold-forward" in ws[0].kernel_info()["checksum_fold"], ws[0].kernel_info()["checksum_fold"]
    # warm the ring: frames 0 .. D saved one by one
    for w in ws:
        for f in range(D + 1):
            w.handle_requests([bg.SaveGameState(f), bg.AdvanceFrame((0,))])
    # the steady tick, pre-marshalled: [Load(F-D), Adv, (Save, Adv) x D]; frames patched per tick
    def template(w):
        reqs = [bg.LoadGameState(0), bg.AdvanceFrame((0,))]
        for i in range(1, D + 1): reqs += [bg.SaveGameState(i), bg.AdvanceFrame((0,))]
        arr, keep, ns = w.build_requests(reqs)
        R = type(arr[0]); sz = C.sizeof(R)
        fv = np.ndarray((len(reqs),), dtype=np.int32, buffer=arr, offset=R.frame.offset, strides=(sz,))
        idx = np.array([0] + [2 * i for i in range(1, D + 1)], dtype=np.intp)
        rel = np.array([0] + list(range(1, D + 1)), dtype=np.int32)
        out = (C.c_uint64 * (2 * ns))()
        return arr, keep, len(reqs), ns, fv, idx, rel, out, np.frombuffer(out, dtype=np.uint64)
    T = [template(w) for w in ws]
    first = {}                                   # frame -> checksum the first time it was saved (SyncTest's own comparison)
    F = D + 1
    t0 = time.perf_counter()
    bad = None
    inflight = []
    for t in range(ticks):
        got = []
        for w, (arr, keep, nreq, ns, fv, idx, rel, out, out_np) in zip(ws, T):
            fv[idx] = rel + (F - D)
            if sync: w.handle_requests_raw(arr, nreq, out); got.append(out_np.copy())
            else: w.enqueue_requests_raw(arr, nreq)
        inflight.append(F)
        F += 1
        if sync or len(inflight) > 1:
            f0 = inflight.pop(0)
            for w, (arr, keep, nreq, ns, fv, idx, rel, out, out_np) in zip(ws, T):
                if sync: break
                w.collect_checksums_raw(out, ns)
                got.append(out_np.copy())
            if not np.array_equal(got[0], got[1]):
                bad = {"tick": t, "why": "fold-forward world and host-fold shadow disagree", "frame": f0}; break
            for k in range(D):
                fr = f0 - D + 1 + k
                c = (int(got[0][2 * k]), int(got[0][2 * k + 1]))
                if first.setdefault(fr, c) != c:
                    bad = {"tick": t, "why": "a re-simulated frame's checksum changed", "frame": fr}; break
            if bad: break
            if len(first) > 64:
                for fr in [x for x in first if x < f0 - 2 * D]: del first[fr]
    for w in ws:
        while w.pending_batches(): w.collect_checksums(64)
    secs = time.perf_counter() - t0
    print(json.dumps({"entities": n, "ticks": ticks if not bad else bad["tick"], "seconds": round(secs, 1), "us_per_tick_both_worlds": round(secs / max(1, t + 1) * 1e6, 2),
                      "equal": bad is None, "first_mismatch": bad, "api": "blocking handle_requests (self-fold vs k_gen_finalize)" if sync else "enqueue / collect, one tick in flight", "checksum_fold": ws[0].kernel_info()["checksum_fold"][:60], "shadow": "host fold (GGRS_FOLD_FORWARD_MIN_WGS=1000000)"}))
    sys.exit(0 if bad is None else 1)


if __name__ == "__main__":
    main()
