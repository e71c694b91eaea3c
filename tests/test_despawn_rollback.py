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
from bevy_ggrs_amd.session import MismatchedChecksum
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
@pytest.mark.parametrize("flags", [0, bg.GGRS_WORLD_NO_GROUPS])     # generic fused groups (markers staged in LDS) / one launch per request
def test_gpu_scripted_matches_oracle(flags):
    got, want = scripted(bg.World(512, max_depth=8, flags=flags)), scripted(OracleWorld(512, 8))
    check_scripted(got)
    for (ka, sa), (kb, sb) in zip(got, want):
        assert ka == kb
        cm.assert_states_equal(sa, sb, ka)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, bg.GGRS_WORLD_NO_GROUPS])
@pytest.mark.parametrize("n,cd", [(200, 3), (5000, 2), (70_000, 4), (3000, 7)])
def test_gpu_synctest_despawn_rollback_matches_oracle(n, cd, flags):
    cap = n + 100
    cs_g, tr_g = synctest_run(bg.World(cap, max_depth=8, flags=flags), n, 14, cd)
    cs_o, tr_o = synctest_run(OracleWorld(cap, 8), n, 14, cd)
    assert cs_g == cs_o
    for t, (a, b) in enumerate(zip(tr_g, tr_o)):
        cm.assert_states_equal(a, b, f"tick {t}")


def particles_with_host_despawn(world, n=3000, ticks=12):
    """Host-issued despawn_rollback() OUTSIDE the GgrsSchedule on the particles world (fused
    request-group path next to live-only marker state).  The command is not part of the rolled-back
    simulation, so the resimulation of its frame resurrects the entities (despawn.rs:69-87) and does
    not despawn them again: SyncTest must report the mismatch (schedule_systems.rs:105-114)."""
    ids = cm.build_particles(world)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, 3)
    trace, err = [], None
    for t in range(ticks):
        if t in (2, 5):
            for slot in range(10 * t, 10 * t + 5):
                world.despawn_rollback(slot)
        try:
            drv.tick((0,))
        except MismatchedChecksum as e:
            err = (t, e.current_frame, tuple(e.mismatched_frames))
            break
        trace.append(state(world, ids))
    return drv.all_checksums, trace, err


def particles_markers_scripted(world, n=3000):
    """Same world, request lists whose rollbacks never cross the marked frame: deterministic."""
    ids = cm.build_particles(world)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    world.set_depth(8)
    world.set_confirmed(0)
    out, cs = [], []
    cs += world.handle_requests([bg.SaveGameState(0)] + [bg.AdvanceFrame((0,))] * 5)
    for slot in range(40, 52):
        world.despawn_rollback(slot)                     # marked with frame 5 (unconfirmed)
    out.append(state(world, ids))
    cs += world.handle_requests([bg.SaveGameState(5), bg.AdvanceFrame((0,)), bg.SaveGameState(6), bg.AdvanceFrame((0,))])
    out.append(state(world, ids))
    # back to 5: marks are not > 5, nobody is resurrected; the resimulated frame 6 must hash as before
    cs += world.handle_requests([bg.LoadGameState(5), bg.AdvanceFrame((0,)), bg.SaveGameState(6), bg.AdvanceFrame((0,)), bg.SaveGameState(7)])
    out.append(state(world, ids))
    # back to 0 (< 5): everybody marked is resurrected and, without the command, stays alive
    cs += world.handle_requests([bg.LoadGameState(0)] + [bg.AdvanceFrame((0,))] * 5 + [bg.SaveGameState(5)])
    out.append(state(world, ids))
    world.set_confirmed(5)
    cs += world.handle_requests([bg.AdvanceFrame((0,)), bg.SaveGameState(6)])
    out.append(state(world, ids))
    return cs, out


def check_markers_scripted(cs, out):
    assert cs[2] == cs[3]                                # Save(6) before and after Load(5)
    assert cs[1] != cs[5]                                # Save(5) with / without the host's despawns
    assert out[0]["disabled"][40:52].all() and not out[0]["alive"][40:52].any()
    assert (out[0]["dframe"][40:52] == 5).all()
    assert out[2]["disabled"][40:52].all()
    assert not out[3]["disabled"].any()                  # resurrected by Load(0) ...
    ttl0 = 1 + (np.arange(3000) % 300)
    assert np.array_equal(out[3]["alive"][40:52], ttl0[40:52] > 5)   # ... and alive unless their Ttl ran out


def test_oracle_host_despawn_outside_schedule_is_a_synctest_mismatch():
    cs, trace, err = particles_with_host_despawn(OracleWorld(3064, 8))
    # despawned on frame 2 (tick 2); frames 2 and 3 are first resimulated by tick 4, checked at tick 5
    assert err == (5, 5, (2, 3))
    assert len(trace) == 5


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_particles_markers_scripted(mode):
    cs, out = particles_markers_scripted(OracleWorld(3064, 8, mode))
    check_markers_scripted(cs, out)


@pytest.mark.gpu
def test_gpu_particles_world_with_markers():
    """The fused request-group path (k_tick) with live-only marker state next to it."""
    got = particles_with_host_despawn(bg.World(3064, max_depth=8))
    want = particles_with_host_despawn(OracleWorld(3064, 8))
    assert got[2] == want[2] == (5, 5, (2, 3))
    assert got[0] == want[0]
    for t, (a, b) in enumerate(zip(got[1], want[1])):
        cm.assert_states_equal(a, b, f"tick {t}")
    cs_g, out_g = particles_markers_scripted(bg.World(3064, max_depth=8))
    cs_o, out_o = particles_markers_scripted(OracleWorld(3064, 8))
    check_markers_scripted(cs_g, out_g)
    assert cs_g == cs_o
    for t, (a, b) in enumerate(zip(out_g, out_o)):
        cm.assert_states_equal(a, b, f"step {t}")


@pytest.mark.gpu
def test_gpu_despawn_rollback_world_runs_as_fused_groups():
    """A world with a despawn_rollback() system is served by k_tick_gen (markers staged in LDS): its ticks are group
    launches, not one launch per request."""
    w = bg.World(400, max_depth=8)
    build(w, 300)
    w.profile_enable(True)
    drv = cm.SyncTestDriver(w, 3)
    for _ in range(10):
        drv.tick((0,))
    prof = w.profile_read()
    assert prof["tick"][1] >= 10 and prof["advance"][1] == 0 and prof["save"][1] == 0
