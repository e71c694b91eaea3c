"""Where does a P2P-shaped tick's host time go when every rollback length has its own specialised kernel?  (profiles/r03zi)
Per-tick enqueue / collect wall times at 100 k entities: random rollback lengths, one fixed length, two alternating lengths.
usage: python scripts/p2p_diag.py [n]     (knobs from the environment: GGRS_JIT_SPECIALISE_AFTER, GGRS_EVENT_ON_KERNEL ...)"""
import ctypes as C
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bevy_ggrs_amd as bg      # noqa: E402
import common as cm             # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    R = 8
    w = bg.World(n, max_depth=R + 1)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(R); w.set_synctest_check_distance(-1)
    lists = {}
    for r in range(R):
        reqs = ([bg.LoadGameState(0)] + [x for i in range(r) for x in (([bg.SaveGameState(0)] if i else []) + [bg.AdvanceFrame((0,))])]) if r else []
        reqs += [bg.SaveGameState(0), bg.AdvanceFrame((0,))]
        arr, keep, n_save = w.build_requests(reqs)
        lists[r] = (arr, keep, n_save, len(reqs), [i for i, q in enumerate(reqs) if isinstance(q, bg.SaveGameState)])
    out = (C.c_uint64 * (2 * R))()
    rng = np.random.default_rng(4)
    F = 0
    pending = []

    def enqueue(r):
        nonlocal F
        r = min(r, F, R - 1)
        arr, _k, n_save, n_req, save_idx = lists[r]
        if r: arr[0].frame = F - r
        for k, i in enumerate(save_idx): arr[i].frame = F - r + (k + 1 if r else 0) if r else F
        if r: arr[save_idx[-1]].frame = F
        if F - R >= 0: w.set_confirmed(F - R)
        t0 = time.perf_counter()
        w.enqueue_requests_raw(arr, n_req)
        dt = time.perf_counter() - t0
        pending.append(n_save); F += 1
        return dt * 1e6, r

    def collect():
        t0 = time.perf_counter()
        w.collect_checksums_raw(out, pending.pop(0))
        return (time.perf_counter() - t0) * 1e6

    def phase(name, pick, ticks):
        w.synchronize()
        rows = []
        t_start = time.perf_counter()
        e, r = enqueue(pick())
        last = time.perf_counter()
        for _ in range(ticks - 1):
            e2, r2 = enqueue(pick())
            c = collect()
            now = time.perf_counter()
            rows.append((e, c, (now - last) * 1e6, r)); last = now; e, r = e2, r2
        c = collect(); rows.append((e, c, (time.perf_counter() - last) * 1e6, r))
        w.synchronize()
        total = (time.perf_counter() - t_start) * 1e6 / ticks
        a = np.array([(x[0], x[1], x[2]) for x in rows])
        med = np.median(a, axis=0); p95 = np.percentile(a, 95, axis=0); mx = a.max(axis=0)
        worst = sorted(range(len(rows)), key=lambda i: -rows[i][2])[:4]
        print(f"{name:34s} {total:7.1f} us/tick | enqueue med {med[0]:6.1f} p95 {p95[0]:6.1f} max {mx[0]:7.1f} | collect med {med[1]:6.1f} p95 {p95[1]:6.1f} max {mx[1]:7.1f} | "
              f"interval med {med[2]:6.1f} | worst ticks {[(i, rows[i][3], round(rows[i][2])) for i in worst]}", flush=True)

    rand = lambda: int(rng.integers(0, R + 1))   # noqa: E731
    if os.environ.get("DIAG_TORCH"):
        import torch
        torch.cuda.synchronize()
    if os.environ.get("DIAG_SETTLE"):             # bench.py's way: short rounds, wait for the worker's build after each
        t0 = time.perf_counter()
        for _ in range(24):
            for _ in range(60):
                enqueue(rand()); collect()
            w.specialise_wait()
        print(f"settled in {(time.perf_counter() - t0) * 1e3:.0f} ms:", w.kernel_info().get("specialised_kernel"), flush=True)
    print("knobs:", {k: v for k, v in os.environ.items() if k.startswith(("GGRS_", "DIAG_"))}, flush=True)
    for rep in range(4):
        phase(f"random lengths #{rep}", rand, 200)
        print("   ", w.kernel_info().get("specialised_kernel"), flush=True)
    w.specialise_wait()
    phase("random lengths (after wait)", rand, 200)
    phase("random lengths (after wait) 2", rand, 200)
    print("   ", w.kernel_info().get("specialised_kernel"), flush=True)
    phase("fixed length 3", lambda: 3, 200)
    phase("fixed length 7", lambda: 7, 200)
    flip = [0]
    def alt():                                    # noqa: E306
        flip[0] ^= 1
        return 3 if flip[0] else 5
    phase("alternating 3 / 5", alt, 200)
    phase("random lengths again", rand, 200)
    w.profile_enable(True)
    for _ in range(50):
        enqueue(rand()); collect()
    print("    kernel by events:", w.profile_read(), flush=True)
    w.profile_enable(False)
    w.close()


if __name__ == "__main__":
    main()
