"""The oracle checked against itself before it is trusted: FLAT vs REFSHAPED storage (same
observable behaviour) and both against the independent numpy twin (oracle/oracle_np.py)."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from bevy_ggrs_amd.session import MismatchedChecksum
from oracle import oracle_np as onp
from oracle.binding import FLAT, REFSHAPED, OracleWorld


def _numpy_particles_checksum(vel, ttl_arr, frames, fps=60):
    """Independent numpy run of `frames` advances from the synthetic start; returns the frame
    checksum (component parts for Velocity and Transform.translation + entity part)."""
    n = vel.shape[0]
    t = [np.zeros(n, np.float32) for _ in range(3)]
    v = [vel[:, 0].copy(), vel[:, 1].copy(), np.zeros(n, np.float32)]
    ttl = ttl_arr.copy()
    alive = np.ones(n, bool)
    for f in range(1, frames + 1):
        dt = onp.dt_bits(fps, f)
        idx = np.nonzero(alive)[0]
        sub_t = [c[idx] for c in t]; sub_v = [c[idx] for c in v]
        onp.np_particles_update(*sub_t, *sub_v, dt)
        for k in range(3):
            t[k][idx] = sub_t[k]; v[k][idx] = sub_v[k]
        ttl[idx] -= np.uint64(1)
        alive[idx[ttl[idx] == 0]] = False
    idx = np.nonzero(alive)[0].astype(np.uint64)
    cs = onp.np_component_checksum(idx, [c[alive].view(np.uint32) for c in v])
    cs ^= onp.np_component_checksum(idx, [c[alive].view(np.uint32) for c in t])
    cs ^= onp.entity_checksum(int(alive.sum()), n)
    return cs, t, v, ttl, alive


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_particles_matches_numpy_twin(mode):
    n = 3000
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    w = OracleWorld(n, 8, mode)
    ids = cm.build_particles(w)
    cm.spawn_particles(w, ids, n, vel, ttl)
    frames = 37
    for _ in range(frames):
        w.advance()
    got = w.save()
    want, t, v, ttl_f, alive = _numpy_particles_checksum(vel, ttl, frames)
    assert got == want
    st = cm.snapshot_state(w, ids)
    assert np.array_equal(st["alive"], alive)
    T, V, L = ids
    for k in range(3):
        assert np.array_equal(st[f"c{T}w{k}"], np.where(alive, t[k].view(np.uint32), 0))
        assert np.array_equal(st[f"c{V}w{k}"], np.where(alive, v[k].view(np.uint32), 0))
    assert np.array_equal(st[f"c{L}w0"], np.where(alive, ttl_f, 0))


def _run_synctest(mode, n, cd, ticks, with_spawn=True):
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    w = OracleWorld(n + 100 * ticks, 16, mode)
    ids = cm.build_particles(w, with_spawn=with_spawn, ttl_init=40)
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, cd)
    fn = cm.frame_spawn_fn()
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if (with_spawn and t % 3 == 1) else 0,), spawn_fn=fn)
    return drv.all_checksums, cm.snapshot_state(w, ids), w


@pytest.mark.parametrize("cd", [1, 2, 7])
def test_flat_equals_refshaped_under_synctest(cd):
    a = _run_synctest(FLAT, 1500, cd, 30)
    b = _run_synctest(REFSHAPED, 1500, cd, 30)
    assert a[0] == b[0]
    cm.assert_states_equal(a[1], b[1], f"cd={cd}")
    # resimulated frames reproduce their first checksum (no MismatchedChecksum was raised) and
    # every frame was saved more than once after warm-up
    frames = [f for f, _ in a[0]]
    assert frames.count(10) == cd


def test_synctest_request_shape():
    """[Load(F-d), Adv, (Save,Adv)x(d-1), Save(F), Adv] after warm-up."""
    from bevy_ggrs_amd.session import SyncTestSession
    s = SyncTestSession(1, 3, 8)
    shapes = []
    for _ in range(8):
        s.add_local_input(0, 0)
        reqs = s.advance_frame()
        s.record_checksums([0] * sum(isinstance(r, bg.SaveGameState) for r in reqs))
        shapes.append([type(r).__name__[0] + (str(r.frame) if not isinstance(r, bg.AdvanceFrame) else "") for r in reqs])
    assert shapes[0] == ["S0", "A"] and shapes[3] == ["S3", "A"]
    assert shapes[4] == ["L1", "A", "S2", "A", "S3", "A", "S4", "A"]
    assert shapes[7] == ["L4", "A", "S5", "A", "S6", "A", "S7", "A"]


def test_mismatch_fires_on_non_determinism():
    """tests/synctest.rs:84-125: a value that is not rolled back changes the resimulated checksum."""
    w = OracleWorld(16, 8, FLAT)
    c = w.register_component("Counter", 4, 1)
    w.checksum_component(c, [0])
    w.spawn(1, {c: [np.zeros(1, np.uint32)]})
    drv = cm.SyncTestDriver(w, 2)
    call_count = 0
    with pytest.raises(MismatchedChecksum):
        for _ in range(10):
            # game logic writing a global (never rolled back) counter into the component:
            # emulate by uploading before each tick's final advance -> resim sees other values
            call_count += 1
            w.upload_word(c, 0, 0, np.array([call_count], np.uint32))
            drv.tick((0,))


def test_confirmed_frame_pruning():
    """tests/synctest.rs:130-153: ConfirmedFrameCount advances, frame-0 snapshot is pruned."""
    w = OracleWorld(16, 8, FLAT)
    c = w.register_component("FrameCounter", 4, 1)
    w.add_system(bg.SYS_ADD_U32, comp=(c,), word=(0,), iparam=(1,))
    w.spawn(1, {c: [np.zeros(1, np.uint32)]})
    drv = cm.SyncTestDriver(w, 5)
    for _ in range(20):
        drv.tick((0,))
    assert not w.has_snapshot(0)
    assert w.has_snapshot(w.frame - 1)
    # final value == frame count (tests/component_rollback.rs:52-65): net one advance per tick
    assert int(w.download_word(c, 0)[0]) == w.frame == 20


def test_despawn_and_rollback():
    """tests/synctest.rs:60-75: Health 10 -> despawn at frame 10, rollback across it, gone after 60."""
    w = OracleWorld(16, 8, FLAT)
    h = w.register_component("Health", 4, 1)
    w.add_system(bg.SYS_SAT_SUB_DESPAWN, comp=(h,), word=(0,), iparam=(1,))
    w.spawn(1, {h: [np.full(1, 10, np.uint32)]})
    drv = cm.SyncTestDriver(w, 5)
    for t in range(60):
        drv.tick((0,))
        if w.frame == 9: assert w.active_count() == 1
    assert w.active_count() == 0


def test_load_missing_frame_is_an_error():
    w = OracleWorld(16, 8, FLAT)
    w.register_component("X", 4, 1)
    w.save()
    with pytest.raises(bg.GgrsHipError) as e:
        w.load(99)
    assert e.value.code == bg.GGRS_E_NO_SNAPSHOT


def _p2p_run(world, n, ticks, rate=40):
    ids = cm.build_particles(world, with_spawn=True, ttl_init=30)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.P2PShapeDriver(world, 8, inputs=lambda f: (cm.INPUT_SPAWN if f % 5 == 2 else 0, 0), spawn_fn=cm.frame_spawn_fn(rate))
    for _ in range(ticks):
        drv.tick()
    return drv, cm.snapshot_state(world, ids)


def test_p2p_shaped_rollbacks_are_deterministic_and_shapes_agree():
    """BASELINE config 4's request shape (variable-depth rollbacks, trailing confirmed frame) on the oracle: a
    frame's checksum never changes when it is resimulated, and both storage shapes agree."""
    a, sa = _p2p_run(OracleWorld(4000, 8, FLAT), 600, 60)
    b, sb = _p2p_run(OracleWorld(4000, 8, REFSHAPED), 600, 60)
    assert a.all_checksums == b.all_checksums and a.depths == b.depths
    assert max(a.depths) == 7 and min(a.depths) == 0
    seen = {}
    for f, c in a.all_checksums:
        assert seen.setdefault(f, c) == c, f"frame {f} changed under resimulation"
    cm.assert_states_equal(sa, sb, "p2p shape")
    # tests/p2p.rs:262-321 (p2p_confirmed_frame_advances_and_prunes_snapshots): once ConfirmedFrameCount advances,
    # snapshots older than it are pruned -- frame 0 is gone, at most max_prediction snapshots remain
    for d in (a, b):
        assert not d.world.has_snapshot(0) and d.world.has_snapshot(d.frame - 1) and d.world.snapshot_count() <= 8


def test_replay_synctest_equals_the_driver_and_per_component_threads_change_nothing():
    """bench.py's parity leg replays the timed ticks through gor_replay_synctest: it must produce exactly the checksums the
    Python SyncTest driver gets request by request (same frames, same order), in both storage shapes, and the REFSHAPED
    variant's one-thread-per-component mode (the 3-thread CPU figure of SURVEY 8d) must not change a bit."""
    from oracle.binding import REFSHAPED, lib
    n, d, ticks = 3000, 4, 14
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ref = None
    for mode, threads in ((FLAT, 1), (REFSHAPED, 1), (REFSHAPED, 3)):
        w = OracleWorld(n, d + 1, mode)
        ids = cm.build_particles(w)
        cm.spawn_particles(w, ids, n, vel, ttl)
        w.set_depth(d + 1)
        lib.gor_set_ref_component_threads(threads)
        try:
            _, cs = w.replay_synctest(d, ticks, 2)
        finally:
            lib.gor_set_ref_component_threads(1)
        st = cm.snapshot_state(w, ids)
        if ref is None:
            w2 = OracleWorld(n, d + 1, FLAT)
            ids2 = cm.build_particles(w2)
            cm.spawn_particles(w2, ids2, n, vel, ttl)
            drv = cm.SyncTestDriver(w2, d, max_prediction=d + 1)
            for _ in range(ticks):
                drv.tick((0,))
            ref = ([c for _, c in drv.all_checksums], cm.snapshot_state(w2, ids2))
            assert len(ref[0]) == (d + 1) + (ticks - d - 1) * d
        assert cs == ref[0], (mode, threads)
        cm.assert_states_equal(st, ref[1], f"replay mode={mode} threads={threads}")
