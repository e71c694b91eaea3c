#!/usr/bin/env python
"""ggrs_hip_fanout_adopt A/B on ONE GPU: what the non-owning ranks pay to follow an adoption -- re-simulating the k adopted frames with the confirmed inputs
(GGRS_ADOPT_RECOMPUTE) against receiving the owner's packed block by one broadcast (GGRS_ADOPT_BROADCAST) -- at world size 2 over the shared-memory stand-in
for RCCL (tests/cpp/rccl_double.cpp: both ranks share the device, so the broadcast is a device-to-device copy through host-shared staging, NOT an xGMI
transfer; the xGMI figure is state_bytes / 153 GB/s per link + the collective's latency).  usage: adopt_ab.py [entities] [frames_ahead]
Prints one JSON line: milliseconds per adopt call on the owner and on the follower, per mode."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_main(rank, size, id_q, q, n, k, rounds):
    try:
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout
        if rank == 0:
            idb = RcclFanout.unique_id()
            for _ in range(size - 1): id_q.put(idb)
        else:
            idb = id_q.get(timeout=120)

        class _Dist:
            def get_rank(self): return rank
            def get_world_size(self): return size
        D = 8
        w = bg.World(n + 64, max_depth=D + 2, device=0)
        ids = cm.build_particles(w)
        if rank == 0:
            vel, ttl = cm.synthetic_particles(n, ttl="throughput")
            cm.spawn_particles(w, ids, n, vel, ttl)
        else:
            w.spawn(0, {})
        native = RcclFanout(w, rank, size, idb)
        zero = lambda b, f: 0
        fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=2, native=native, branch_input=zero, confirmed_input=lambda f: 0, retain="all")
        out = {}
        for mode in ("recompute", "broadcast"):
            ts = []
            for r in range(rounds):
                fan.step()
                w.synchronize()
                t0 = time.perf_counter()
                fan.adopt(0, k, broadcast=(mode == "broadcast"))          # branch 0 lives on rank 0: rank 1 follows
                w.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            out[mode] = {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4)}
        cs = w.save()
        native.close()
        q.put((rank, "ok", out, cs, w.state_bytes()))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}", traceback.format_exc(), 0))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    import subprocess
    src, lib = os.path.join(ROOT, "tests", "cpp", "rccl_double.cpp"), os.path.join(ROOT, "tests", "cpp", "_build", "librccl_double.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", lib, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    os.environ["GGRS_RCCL_LIB"] = lib
    ctx = mp.get_context("spawn")
    q, id_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, 2, id_q, q, n, k, 12)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs: p.join(30)
    assert all(r[1] == "ok" for r in res), res
    assert res[0][3] == res[1][3], "the ranks diverged"
    print(json.dumps({"entities": n, "frames_ahead": k, "state_bytes": res[0][4], "owner_rank0": res[0][2], "follower_rank1": res[1][2], "ranks_agree": True,
                      "transport": "shared-memory stand-in for RCCL on one GPU (no xGMI)",
                      "xgmi_estimate_ms_broadcast": round(res[0][4] / 153e9 * 1e3, 4)}))


if __name__ == "__main__":
    main()
