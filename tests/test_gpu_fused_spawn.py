"""spawn_particles INSIDE a request group (kernel_gen.hpp "fused spawn"): the reference's spawn system
(examples/stress_tests/particles.rs:254-270) fires through Commands, which Bevy applies at the end of the schedule; rounds 1-3 ended the
request group there and ran the spawn as its own launches, round 4 lets the group's launch append the rows.  Everything below is compared
with the CPU oracle bit for bit -- checksums of every Save, the final live state incl. the presence masks of components the spawn bundle
does NOT carry -- with the spawn fused (the default), on the one-launch-per-request path, and in a world whose schedule holds TWO spawn
systems (the generator fuses one: the firing spawn then ends its group and runs as its own launches, as in rounds 1-3)."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu


def _session(world, n, ticks, D, schema, rate, hold, two_systems=False):
    ids = cm.build_particles(world, with_spawn=True, ttl_init=37, schema=schema)
    if two_systems: world.add_system(bg.SYS_PARTICLES_SPAWN, comp=ids[:3], iparam=(11, 1 << 6))      # a second spawner on another input bit (never held here)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, D, max_prediction=D + 1)
    fn = cm.frame_spawn_fn(rate)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if hold(t) else 0,), spawn_fn=fn)
    return drv.all_checksums, cm.snapshot_state(world, ids)


@pytest.mark.parametrize("mode", ["fused", "per_request", "two_spawn_systems"])
@pytest.mark.parametrize("n,schema,rate,D", [(3000, "headline", 100, 8), (3000, "full", 70, 5), (300_000, "full", 100, 8), (700_000, "headline", 130, 8),
                                             (5000, "allhot", 100, 8), (450_000, "allhot", 100, 8)])
def test_spawn_key_held_matches_the_oracle(n, schema, rate, D, mode, monkeypatch):
    """The stress_test with the spawn key down: every frame of every tick -- resimulated ones included -- spawns `rate` particles, Ttl despawns
    run beside them, the world grows across 64-slot, 256-slot and layout-tile boundaries.  `full`: the spawn bundle carries Transform,
    Velocity and Ttl only, so spawned entities must come out WITHOUT GlobalTransform / the visibility bytes (presence masks)."""
    if mode == "per_request": monkeypatch.setenv("GGRS_TICK_JIT", "0")
    two = mode == "two_spawn_systems"
    ticks = 14
    cap = n + rate * (ticks + 2 * D + 4)
    g = bg.World(cap, max_depth=D + 1)
    a = _session(g, n, ticks, D, schema, rate, lambda t: True, two)
    info = g.kernel_info()
    o = OracleWorld(cap, D + 1, FLAT)
    b = _session(o, n, ticks, D, schema, rate, lambda t: True, two)
    if mode == "per_request": assert info["request_group_kernel"].startswith("per-request"), info
    else: assert info["spawn_system"].startswith("ends the request group" if two else "runs inside"), info
    fuse = mode
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


@pytest.mark.parametrize("stage_floats", [1024, 4096, 1 << 20])
def test_payload_ring_wraps_and_falls_back_under_the_pipelined_api(stage_floats, monkeypatch):
    """Spawn payloads live in a ring that a collected batch frees (host_world.hpp).  With the ring shrunk (GGRS_STAGE_BYTES) a session that
    holds the spawn key through enqueue / collect with one tick in flight wraps it every other tick (16 KiB) or fills it inside one
    list (4 KiB: the enqueue then waits for the stream, starts the ring over -- and forgets which payloads it had staged, ADVICE r4 -- ) --
    the checksums must not care."""
    from bevy_ggrs_amd.session import SyncTestSession
    monkeypatch.setenv("GGRS_STAGE_BYTES", str(4 * stage_floats))
    n, D, rate, ticks = 5000, 8, 100, 40
    cap = n + rate * (ticks + 2 * D + 4)
    fn = cm.frame_spawn_fn(rate)

    def lists():
        sess = SyncTestSession(1, D, D + 1, 0)
        frame = 0
        for _ in range(ticks):
            sess.add_local_input(0, cm.INPUT_SPAWN)
            reqs = sess.advance_frame()
            cur = frame
            for r in reqs:
                if isinstance(r, bg.LoadGameState): cur = r.frame
                elif isinstance(r, bg.AdvanceFrame):
                    r.spawn_vx, r.spawn_vy = fn(cur); cur += 1
            frame += 1
            yield sess, reqs
    g = bg.World(cap, max_depth=D + 1)
    o = OracleWorld(cap, D + 1, FLAT)
    got, want = [], []
    for w in (g, o):
        ids = cm.build_particles(w, with_spawn=True, ttl_init=37)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        w.set_depth(D + 1)
    g.set_synctest_check_distance(D)
    pending = []
    for sess, reqs in lists():                                    # GPU: enqueue tick k+1, then collect tick k
        pending.append(g.enqueue_requests(reqs))
        if len(pending) > 1:
            got += g.collect_checksums(pending.pop(0))
        sess.record_checksums([0] * sum(isinstance(r, bg.SaveGameState) for r in reqs))     # (the session only counts them here)
    while pending: got += g.collect_checksums(pending.pop(0))
    for sess, reqs in lists():                                    # oracle: the same lists, request by request (its confirmed rule applied per request)
        for r in reqs:
            c = o.frame - D
            if c >= 0: o.set_confirmed(c)
            want += o.handle_requests([r])
        sess.record_checksums([0] * sum(isinstance(r, bg.SaveGameState) for r in reqs))
    assert len(got) == len(want) > ticks and got == want
    cm.assert_states_equal(cm.snapshot_state(g, (0, 1, 2)), cm.snapshot_state(o, (0, 1, 2)), f"payload ring {stage_floats}")
