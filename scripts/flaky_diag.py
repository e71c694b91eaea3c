"""Run the GPU tests that precede tests/test_gpu_parity.py in one process, then the flaky case by hand with state dumps."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np
import pytest
prefix = sys.argv[1:] or ["tests/test_box_game.py", "tests/test_cpp_host.py", "tests/test_despawn_rollback.py", "tests/test_gpu_custom_system.py", "tests/test_gpu_gen_groups.py", "tests/test_gpu_golden.py"]
if prefix != ["none"]:
    rc = pytest.main(prefix + ["-m", "gpu", "-q", "-p", "no:cacheprovider"])
    print("prefix rc", rc)
import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld


def scenario(tag):
    n, cd = 1, 2
    cap = n + 100 * 12 + 64
    g, o = bg.World(cap, max_depth=16), OracleWorld(cap, 16, FLAT)
    ws = []
    for w in (g, o):
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        ids = cm.build_particles(w, with_spawn=True, ttl_init=40)
        cm.spawn_particles(w, ids, n, vel, ttl)
        ws.append((w, ids, cm.SyncTestDriver(w, cd)))
    fn = cm.frame_spawn_fn(100)
    for t in range(6):
        cs = []
        for w, ids, drv in ws:
            cs.append(drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn))
        sg, so = cm.snapshot_state(ws[0][0], ws[0][1]), cm.snapshot_state(ws[1][0], ws[1][1])
        diff = [k for k in so if not np.array_equal(np.asarray(sg[k]), np.asarray(so[k]))]
        print(f"[{tag}] tick {t}: checksums equal {cs[0] == cs[1]} {[hex(c) for c in cs[0]]} vs {[hex(c) for c in cs[1]]}; state diff keys {diff}")
        for k in diff[:6]:
            a, b = np.asarray(sg[k]), np.asarray(so[k])
            if a.shape == b.shape and a.ndim:
                idx = np.nonzero(a != b)[0]
                print(f"    {k}: {idx.size} slots differ, first {idx[:8]}, gpu {a[idx[:4]]}, oracle {b[idx[:4]]}")
            else:
                print(f"    {k}: gpu {a} oracle {b}")


def exact(tag):
    """the test's own flow: all ticks on the GPU world, then all on the oracle, compare at the end"""
    n, cd, ticks = 1, 2, 12
    cap = n + 100 * ticks + 64
    out = []
    for w in (bg.World(cap, max_depth=16), OracleWorld(cap, 16, FLAT)):
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        ids = cm.build_particles(w, with_spawn=True, ttl_init=40)
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, cd)
        fn = cm.frame_spawn_fn(100)
        for t in range(ticks):
            drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
        out.append((drv.all_checksums, cm.snapshot_state(w, ids)))
    bad = [(i, fa, hex(ca), hex(cb)) for i, ((fa, ca), (fb, cb)) in enumerate(zip(out[0][0], out[1][0])) if ca != cb]
    diff = [k for k in out[1][1] if not np.array_equal(np.asarray(out[0][1][k]), np.asarray(out[1][1][k]))]
    print(f"[{tag}] {len(bad)} of {len(out[0][0])} checksums differ: {bad[:12]}; final state diff keys {diff}")


exact("exact1")
exact("exact2")
scenario("first")
scenario("second")
