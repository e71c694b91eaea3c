"""GGRS_WORLD_CONTIG_ARENA (include/ggrs_hip.h): the opt-in physically contiguous arena.  Its precondition -- the world is
created before the process has freed device memory -- is a property of the PROCESS, so every case runs in a fresh one:
the headline world and two mid-size worlds on a contiguous arena against the CPU oracle, the library's safety net (the flag
is ignored once the process has freed a paged arena of its own), and the sequence that used to break later worlds (a contiguous
world closed, a paged world opened next)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld, lib as olib
olib.gor_set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
n, ticks, free_paged_first = {n}, {ticks}, {free_first}
if free_paged_first:                      # a paged world lives and dies first: the later request must be ignored
    p = bg.World(200_000, max_depth=4); ids = cm.build_particles(p); vel, ttl = cm.synthetic_particles(1000); cm.spawn_particles(p, ids, 1000, vel, ttl); p.save(); p.close()
vel, ttl = cm.synthetic_particles(n, ttl="despawn")
res, info = [], None
for w in (bg.World(n, max_depth=9, flags=bg.GGRS_WORLD_CONTIG_ARENA), OracleWorld(n, 9, FLAT)):
    ids = cm.build_particles(w)
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, 8, max_prediction=9)
    for _ in range(ticks): drv.tick((0,))
    res.append(drv.all_checksums)
    if info is None: info = w.kernel_info()
    w.close()
print("RESULT " + json.dumps({{"equal": res[0] == res[1], "saves": len(res[0]), "arena": info["arena"], "kernel": info["request_group_kernel"]}}))
'''


def _child(n, ticks, free_first=False):
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, n=n, ticks=ticks, free_first=free_first)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[-1][7:])


@pytest.mark.parametrize("n,ticks", [(1_000_000, 12), (450_000, 11), (700_000, 11)])
def test_contiguous_arena_world_matches_oracle(n, ticks):
    out = _child(n, ticks)
    assert out["arena"].startswith("contiguous"), out          # the request was honoured: first allocation of the process, k_tick3 world, < 1.5 GiB
    assert out["kernel"].startswith(("k_tick3", "ggrs_jit_tick")), out
    assert out["equal"] and out["saves"] >= 8 * (ticks - 9), out


def test_flag_is_ignored_after_the_process_freed_a_paged_arena():
    out = _child(450_000, 10, free_first=True)
    assert out["arena"].startswith("paged"), out               # the library's safety net (ggrs_hip.h)
    assert out["equal"], out


CHILD_SEQ = r'''
import gc, json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld
out = []
for flags, n in {seq!r}:
    cap = n + 100 * 30 + 64
    res, info = [], None
    for w in (bg.World(cap, max_depth=16, flags=flags), OracleWorld(cap, 16, FLAT)):
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        ids = cm.build_particles(w, with_spawn=True, ttl_init=40)
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, 8); fn = cm.frame_spawn_fn(100)
        for t in range(30): drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
        res.append((drv.all_checksums, cm.snapshot_state(w, ids)))
        if info is None: info = w.kernel_info()
        w.close()
    state_equal = True
    try: cm.assert_states_equal(res[0][1], res[1][1])
    except AssertionError: state_equal = False
    out.append({{"flags": flags, "arena": info["arena"], "equal": res[0][0] == res[1][0] and state_equal}})
    gc.collect()
print("RESULT " + json.dumps(out))
'''


def test_a_closed_contiguous_world_does_not_break_later_worlds():
    """profiles/r03fc: closing a contiguous world and opening a paged one next (the UNFUSED world: one kernel per system, the most
    launches per tick) failed on 6 of 6 fresh boxes while contiguous arenas were handed back with hipFree -- the later world's
    kernels stopped seeing each other's writes.  Contiguous arenas are parked for the life of the process instead; the second
    contiguous world below reuses the first one's arena."""
    import bevy_ggrs_amd as bg
    C, U, N = bg.GGRS_WORLD_CONTIG_ARENA, bg.GGRS_WORLD_UNFUSED, bg.GGRS_WORLD_NO_GROUPS
    seq = [(C, 10_000), (U, 10_000), (C, 10_000), (N, 10_000), (U, 7_000), (C, 4_000), (U, 10_000)]
    r = subprocess.run([sys.executable, "-c", CHILD_SEQ.format(root=ROOT, seq=seq)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1][7:])
    assert all(o["equal"] for o in out), out
    assert out[0]["arena"].startswith("contiguous") and out[2]["arena"].startswith("contiguous") and out[5]["arena"].startswith("contiguous"), out
    assert out[1]["arena"].startswith("paged"), out
