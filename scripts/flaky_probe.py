"""Repeat one small SyncTest parity case many times and count checksum mismatches against the oracle (race hunting)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld


def run(world, n, cd, ticks):
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ids = cm.build_particles(world, with_spawn=True, ttl_init=40)
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, cd)
    fn = cm.frame_spawn_fn(100)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
    return drv.all_checksums


n, cd, ticks, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cap = n + 100 * ticks + 64
want = run(OracleWorld(cap, 16, FLAT), n, cd, ticks)
bad = 0
first = None
for r in range(reps):
    got = run(bg.World(cap, max_depth=16), n, cd, ticks)
    if got != want:
        bad += 1
        if first is None:
            first = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:3]
print(f"n={n} cd={cd} ticks={ticks}: {bad} / {reps} runs differ from the oracle", first)
