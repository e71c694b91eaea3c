"""spawn_particles INSIDE a request group (kernel_gen.hpp "fused spawn"): the reference's spawn system
(examples/stress_tests/particles.rs:254-270) fires through Commands, which Bevy applies at the end of the schedule; rounds 1-3 ended the
request group there and ran the spawn as its own launches, round 4 lets the group's launch append the rows.  Everything below is compared
with the CPU oracle bit for bit -- checksums of every Save, the final live state incl. the presence masks of components the spawn bundle
does NOT carry -- with the spawn fused (default) and with GGRS_JIT_FUSE_SPAWN=0."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu


def _session(world, n, ticks, D, schema, rate, hold):
    ids = cm.build_particles(world, with_spawn=True, ttl_init=37, schema=schema)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, D, max_prediction=D + 1)
    fn = cm.frame_spawn_fn(rate)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if hold(t) else 0,), spawn_fn=fn)
    return drv.all_checksums, cm.snapshot_state(world, ids)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("n,schema,rate,D", [(3000, "headline", 100, 8), (3000, "full", 70, 5), (300_000, "full", 100, 8), (700_000, "headline", 130, 8)])
def test_spawn_key_held_matches_the_oracle(n, schema, rate, D, fuse, monkeypatch):
    """The stress_test with the spawn key down: every frame of every tick -- resimulated ones included -- spawns `rate` particles, Ttl despawns
    run beside them, the world grows across 64-slot, 256-slot and layout-tile boundaries.  `full`: the spawn bundle carries Transform,
    Velocity and Ttl only, so spawned entities must come out WITHOUT GlobalTransform / the visibility bytes (presence masks)."""
    if not fuse: monkeypatch.setenv("GGRS_JIT_FUSE_SPAWN", "0")
    ticks = 14
    cap = n + rate * (ticks + 2 * D + 4)
    g = bg.World(cap, max_depth=D + 1)
    a = _session(g, n, ticks, D, schema, rate, lambda t: True)
    info = g.kernel_info()
    o = OracleWorld(cap, D + 1, FLAT)
    b = _session(o, n, ticks, D, schema, rate, lambda t: True)
    assert info["spawn_system"].startswith("runs inside" if fuse else "ends the request group"), info
    assert len(a[0]) == len(b[0]) > ticks
    for (fa, ca), (fb, cb) in zip(a[0], b[0]):
        assert fa == fb and ca == cb, f"frame {fa}: gpu {ca:#x} oracle {cb:#x}"
    cm.assert_states_equal(a[1], b[1], f"spawn held n={n} {schema} fuse={fuse}")
    assert a[1]["len"] == n + rate * ticks                       # one net spawn per tick survives the rollbacks


def test_spawns_on_some_frames_and_in_branch_lists():
    """Spawns on every third frame only (groups with and without a spawn step alternate, row masks change from Save to Save inside a group) and
    a branch list whose branches spawn or not by their input byte: the dead-snapshot rule and the batches must still give every branch's
    checksums, and the ring / live state must be the oracle's afterwards."""
    n, D, rate = 20_000, 6, 90
    res = []
    for w in (bg.World(n + rate * 80, max_depth=D + 2), OracleWorld(n + rate * 80, D + 2, FLAT)):
        ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        fn = cm.frame_spawn_fn(rate)
        drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
        for t in range(12):
            drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
        out = list(drv.all_checksums)
        C = w.frame - 1
        w.set_synctest_check_distance(-1); w.set_confirmed(max(0, C - D))
        reqs = []
        for b in range(6):                                        # six branches off the newest snapshot: even ones hold the spawn key
            reqs.append(bg.LoadGameState(C))
            for i in range(4):
                adv = bg.AdvanceFrame((cm.INPUT_SPAWN if b % 2 == 0 else 0,))
                if b % 2 == 0: adv.spawn_vx, adv.spawn_vy = fn(C + i)
                reqs += [adv, bg.SaveGameState(C + 1 + i)]
            reqs.append(bg.AdvanceFrame((0,)))
        out += [(None, c) for c in w.handle_requests(reqs)]
        out += [(None, c) for c in w.handle_requests([bg.SaveGameState(w.frame)])]
        res.append((out, cm.snapshot_state(w, ids)))
    assert res[0][0] == res[1][0]
    cm.assert_states_equal(res[0][1], res[1][1], "branch lists with spawns")
    per = 4
    br = [res[0][0][-(6 * per + 1) + b * per: -(6 * per + 1) + (b + 1) * per] for b in range(6)]
    assert br[0] == br[2] == br[4] and br[1] == br[3] == br[5] and br[0] != br[1]


def test_spawn_beyond_capacity_is_an_error_not_a_corruption():
    n = 1000
    g = bg.World(n + 150, max_depth=4)
    ids = cm.build_particles(g, with_spawn=True)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(g, ids, n, vel, ttl)
    fn = cm.frame_spawn_fn(100)
    a = bg.AdvanceFrame((cm.INPUT_SPAWN,)); a.spawn_vx, a.spawn_vy = fn(0)
    g.handle_requests([bg.SaveGameState(0), a])                   # 1100 <= 1150
    b = bg.AdvanceFrame((cm.INPUT_SPAWN,)); b.spawn_vx, b.spawn_vy = fn(1)
    with pytest.raises(bg.GgrsHipError) as e:
        g.handle_requests([bg.SaveGameState(1), b])               # 1200 > 1150
    assert e.value.code == bg.GGRS_E_CAPACITY and "exceeds capacity" in str(e.value)
    assert g.len == n + 100
