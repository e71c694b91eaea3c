"""The contiguous-arena failure (profiles/r03fc: test_particles_synctest_checksums_and_state[UNFUSED-10000-8-30], 6 of 6 fresh boxes),
taken apart: after every SyncTest tick, read one column back through the C ABI (twice, into differently pre-filled buffers) and
compare it with the CPU oracle's; on the first difference print which slots differ and whether a SECOND read of the same column,
a checksum request and a read after hipDeviceSynchronize agree.  Arms are chosen with the environment (GGRS_ARENA_CONTIG=2 ...).
usage: python scripts/contig_diag.py [flags[,flags...] [n cd ticks]]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bevy_ggrs_amd as bg      # noqa: E402
import common as cm             # noqa: E402
from oracle.binding import FLAT, OracleWorld   # noqa: E402


def ranges(idx):
    if idx.size == 0: return "-"
    cut = np.nonzero(np.diff(idx) != 1)[0]
    starts = np.concatenate(([idx[0]], idx[cut + 1])); ends = np.concatenate((idx[cut], [idx[-1]]))
    return " ".join(f"{a}..{b}" for a, b in list(zip(starts, ends))[:12]) + (f" (+{len(starts) - 12} more)" if len(starts) > 12 else "")


def read(w, cid, k, n, fill):
    _, wb, _ = w._comps[cid]
    out = np.full(n, fill, dtype={4: np.uint32, 8: np.uint64}[wb])
    w._check(w._fn("download_word")(w._p, cid, k, 0, n, out.ctypes.data))
    return out


def main():
    import gc
    # "8,2": a NO_GROUPS world, closed, then an UNFUSED one; "2:100,8,2": a 100-entity UNFUSED world first; DIAG_KEEP=1 keeps every world open
    seq = [(int(x.split(":")[0]), int(x.split(":")[1]) if ":" in x else None) for x in (sys.argv[1] if len(sys.argv) >= 2 else "2").split(",")]
    n, cd, ticks = (int(x) for x in (sys.argv[2:5] if len(sys.argv) >= 5 else (10000, 8, 30)))
    keep = []
    for flags, n_this in seq:
        print(f"--- world flags={flags} n={n_this or n} cd={cd} ticks={ticks}", flush=True)
        w = one_world(flags, n_this or n, cd, ticks)
        if os.environ.get("DIAG_KEEP"): keep.append(w)
        del w
        gc.collect()
    return 0


def one_world(flags, n, cd, ticks):
    cap = n + 100 * ticks + 64
    g, o = bg.World(cap, max_depth=16, flags=flags), OracleWorld(cap, 16, FLAT)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ids_g = cm.build_particles(g, with_spawn=True, ttl_init=40); ids_o = cm.build_particles(o, with_spawn=True, ttl_init=40)
    cm.spawn_particles(g, ids_g, n, vel, ttl); cm.spawn_particles(o, ids_o, n, vel, ttl)
    print("arena:", {k: v for k, v in g.kernel_info().items() if k in ("arena", "generated_kernel", "request_group_kernel", "row_versions")}, flush=True)
    dg, do = cm.SyncTestDriver(g, cd), cm.SyncTestDriver(o, cd)
    fn = cm.frame_spawn_fn(100)
    bad_ticks = 0
    for t in range(ticks):
        inp = (cm.INPUT_SPAWN if t % 3 == 1 else 0,)
        dg.tick(inp, spawn_fn=fn); do.tick(inp, spawn_fn=fn)
        if dg.all_checksums != do.all_checksums and bad_ticks == 0:
            k = next(i for i, (x, y) in enumerate(zip(dg.all_checksums, do.all_checksums)) if x != y)
            print(f"tick {t}: first checksum difference at compared frame {dg.all_checksums[k][0]}: gpu {dg.all_checksums[k][1]:#x} oracle {do.all_checksums[k][1]:#x}", flush=True)
            bad_ticks += 1
        m = g.len
        assert m == o.len
        alive = o.alive_mask(m)
        for cid, k in (() if os.environ.get("DIAG_NO_READS") else ((ids_g[2], 0), (ids_g[0], 0), (ids_g[1], 1))):
            want = np.where(alive, o.download_word(cid, k, 0, m), 0)
            a = np.where(alive, read(g, cid, k, m, 0), 0)
            b = np.where(alive, read(g, cid, k, m, 0xEE), 0)
            da, db = np.nonzero(a != want)[0], np.nonzero(b != want)[0]
            if da.size or db.size:
                bad_ticks += 1
                print(f"tick {t} frame {g.frame} c{cid}w{k}: read#1 differs at {da.size} slots [{ranges(da)}], read#2 at {db.size} [{ranges(db)}]", flush=True)
                if da.size:
                    s = da[:4]; print("   slots", s, "gpu", a[s], "oracle", want[s], "initial ttl", ttl[s] if s.max() < n else "-")
                if bad_ticks > 6: break
        if bad_ticks > 6: break
    same = dg.all_checksums == do.all_checksums
    print(f"checksums equal over {len(dg.all_checksums)} compared frames: {same}; ticks with a stale read: {bad_ticks}")
    cs_g, cs_o = g.save(), o.save()
    print(f"one more SaveWorld: gpu {cs_g:#x} oracle {cs_o:#x} equal={cs_g == cs_o}")
    a = cm.snapshot_state(g, ids_g); b = cm.snapshot_state(o, ids_o)
    try:
        cm.assert_states_equal(a, b, "final"); print("final state: equal")
    except AssertionError as e:
        print("final state:", str(e)[:300])
    return g


if __name__ == "__main__":
    sys.exit(main())
