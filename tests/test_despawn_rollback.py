"""RollbackDespawned deferred despawn + non-rollback components.

Reference: src/snapshot/despawn.rs -- `despawn_rollback()` (:114-143) disables an entity instead of
freeing it while its frame is unconfirmed, `resurrect_entities` (:69-87) re-enables it when LoadWorld
goes back before the marked frame, `despawn_confirmed_entities` (:89-112) frees it once the frame is
confirmed.  The reference ships no test for this module; the scenarios below pin the restated
semantics on the oracle (both storage shapes must agree) and the GPU tests require the HIP engine to
reproduce the oracle bit for bit, including the peer-local marker state.
"""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
from oracle.binding import FLAT, REFSHAPED, OracleWorld

import common as cm


def build(world, n, mode=bg.DESPAWN_ROLLBACK, checksum=True):
    """tests/synctest.rs:26-52 (Health + decrease_health) plus a non-rollback `Mesh` handle column."""
    H = world.register_component("Health", 4, 1)
    M = world.register_component("Mesh", 4, 2, rollback=False)
    if checksum:
        world.checksum_component(H, [0])
    world.add_system(bg.SYS_SAT_SUB_DESPAWN, comp=(H,), word=(0,), iparam=(1, mode))
    health = (1 + (np.arange(n) % 5)).astype(np.uint32)          # dies entering frames 1..5
    mesh = [np.arange(n, dtype=np.uint32) + 1000, np.arange(n, dtype=np.uint32) * 7]
    world.spawn(n, {H: [health], M: mesh})
    return H, M


def state(world, ids):
    s = cm.snapshot_state(world, ids)
    n = world.len
    dis = world.disabled_mask(n)
    s["disabled"] = dis
    s["dframe"] = np.where(dis, world.despawned_frames(0, n), 0)
    return s


def scripted(world, n=300):
    """Save(0); 3 x Advance; Load(0); ... observing the world after every step."""
    H, M = build(world, n)
    ids = (H, M)
    world.set_depth(8)
    world.set_confirmed(0)
    out = []
    world.handle_requests([bg.SaveGameState(0)])
    for f in range(3):                                   # frames 1..3: health 1, 2, 3 die -> disabled
        world.handle_requests([bg.AdvanceFrame((0,))])
        out.append(("adv", state(world, ids)))
    world.handle_requests([bg.SaveGameState(3)])
    world.handle_requests([bg.LoadGameState(0)])         # marks 1..3 > 0: everybody is resurrected
    out.append(("load0", state(world, ids)))
    world.handle_requests([bg.AdvanceFrame((0,)), bg.AdvanceFrame((0,))])     # frames 1, 2 again
    out.append(("resim2", state(world, ids)))
    world.handle_requests([bg.SaveGameState(2)])
    world.handle_requests([bg.LoadGameState(2)])         # marks 1, 2 are not > 2: nobody is resurrected
    out.append(("load2", state(world, ids)))
    world.set_confirmed(1)                               # frame 1 confirmed: its despawns become final
    world.handle_requests([bg.AdvanceFrame((0,))])       # frame 3: DespawnConfirmed runs first
    out.append(("confirmed1", state(world, ids)))
    world.despawn_rollback(7)                            # host-issued command on the unconfirmed frame 3
    out.append(("cmd", state(world, ids)))
    world.set_confirmed(3)
    world.despawn_rollback(8)                            # frame already confirmed: plain despawn
    world.handle_requests([bg.AdvanceFrame((0,))])
    out.append(("confirmed3", state(world, ids)))
    return out


def check_scripted(out, n=300):
    health0 = 1 + (np.arange(n) % 5)
    s = out[2][1]                                        # after 3 advances
    assert s["frame"] == 3
    assert np.array_equal(s["alive"], health0 > 3)
    assert np.array_equal(s["disabled"], health0 <= 3)
    assert np.array_equal(s["dframe"], np.where(health0 <= 3, health0, 0))    # marked with the frame they died in
    s = dict(out)["load0"]
    assert s["alive"].all() and not s["disabled"].any()
    # resurrected entities kept their non-rollback component and its data (that is the point of despawn.rs)
    M = 1
    assert s[f"present{M}"].all()
    assert np.array_equal(s[f"c{M}w0"], np.arange(n) + 1000) and np.array_equal(s[f"c{M}w1"], np.arange(n) * 7)
    assert np.array_equal(s["c0w0"], health0)            # rollback component restored from the snapshot
    s = dict(out)["load2"]
    assert np.array_equal(s["disabled"], health0 <= 2) and np.array_equal(s["alive"], health0 > 2)
    s = dict(out)["confirmed1"]                          # marks <= 1 freed; frame 3's deaths newly marked
    assert np.array_equal(s["disabled"], (health0 == 2) | (health0 == 3))
    assert np.array_equal(s["dframe"], np.where(health0 == 2, 2, np.where(health0 == 3, 3, 0)))
    assert not s[f"present{M}"][health0 == 1].any()      # a freed entity's non-rollback component is gone
    s = dict(out)["cmd"]
    assert s["disabled"][7] and s["dframe"][7] == 3 and not s["alive"][7]
    s = dict(out)["confirmed3"]
    assert not s["alive"][8] and not s["disabled"][8]    # plainly despawned (its frame was confirmed)
    want = health0 == 4                                  # everything marked <= 3 is freed; frame 4's deaths are marked
    want[8] = False
    assert np.array_equal(s["disabled"], want)


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_scripted_semantics(mode):
    check_scripted(scripted(OracleWorld(512, 8, mode)))


def test_oracle_shapes_agree():
    a, b = scripted(OracleWorld(512, 8, FLAT)), scripted(OracleWorld(512, 8, REFSHAPED))
    for (ka, sa), (kb, sb) in zip(a, b):
        assert ka == kb
        cm.assert_states_equal(sa, sb, ka)


def test_immediate_despawn_loses_non_rollback_component_on_reload():
    """The behaviour RollbackDespawned exists to avoid: a plainly despawned entity that LoadWorld has to
    re-create (entity.rs:80-90) comes back with its rollback components only."""
    w = OracleWorld(64, 8)
    H, M = build(w, 10, mode=bg.DESPAWN_IMMEDIATE)
    w.set_depth(8); w.set_confirmed(0)
    w.handle_requests([bg.SaveGameState(0), bg.AdvanceFrame((0,)), bg.LoadGameState(0)])
    s = cm.snapshot_state(w, (H, M))
    assert s["alive"].all()
    died = (1 + np.arange(10) % 5) == 1
    assert np.array_equal(s[f"present{M}"], ~died)
    assert np.array_equal(s[f"present{H}"], np.ones(10, bool))


def synctest_run(world, n, ticks, cd):
    H, M = build(world, n)
    drv = cm.SyncTestDriver(world, cd)
    trace = []
    for _ in range(ticks):
        drv.tick((0,))
        trace.append(state(world, (H, M)))
    return drv.all_checksums, trace


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_synctest_despawn_rollback(mode):
    """tests/synctest.rs:60-75 with despawn_rollback: no mismatch, entity confirmed gone."""
    cs, trace = synctest_run(OracleWorld(256, 8, mode), 200, 16, 3)
    seen = {}
    for f, c in cs:
        assert seen.setdefault(f, c) == c, f"checksum mismatch at frame {f}"
    last = trace[-1]
    assert not last["alive"].any() and not last["disabled"].any()
    assert not last["present1"].any()


@pytest.mark.gpu
def test_gpu_scripted_matches_oracle():
    got, want = scripted(bg.World(512, max_depth=8)), scripted(OracleWorld(512, 8))
    check_scripted(got)
    for (ka, sa), (kb, sb) in zip(got, want):
        assert ka == kb
        cm.assert_states_equal(sa, sb, ka)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cd", [(200, 3), (5000, 2), (70_000, 4)])
def test_gpu_synctest_despawn_rollback_matches_oracle(n, cd):
    cap = n + 100
    cs_g, tr_g = synctest_run(bg.World(cap, max_depth=8), n, 14, cd)
    cs_o, tr_o = synctest_run(OracleWorld(cap, 8), n, 14, cd)
    assert cs_g == cs_o
    for t, (a, b) in enumerate(zip(tr_g, tr_o)):
        cm.assert_states_equal(a, b, f"tick {t}")


@pytest.mark.gpu
def test_gpu_particles_world_with_markers():
    """The fused request-group path (k_tick) with live-only marker state next to it: host-issued
    despawn_rollback on the particles world, rolled back and resimulated by SyncTest."""
    n = 3000
    res = []
    for w in (bg.World(n + 64, max_depth=8), OracleWorld(n + 64, 8)):
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, 3)
        trace = []
        for t in range(12):
            if t in (2, 5):
                for slot in range(10 * t, 10 * t + 5):
                    w.despawn_rollback(slot)
            drv.tick((0,))
            trace.append(state(w, ids))
        res.append((drv.all_checksums, trace))
    assert res[0][0] == res[1][0]
    for t, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        cm.assert_states_equal(a, b, f"tick {t}")
