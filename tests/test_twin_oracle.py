"""The C++ oracle (oracle/ggrs_oracle.cpp, the checker of every GPU parity test) against the independently written
numpy twin of the WHOLE tick (oracle/twin_np.py: dict-of-RollbackId snapshots, deque ring, set-algebra entity
reconcile, despawn markers), driven through the same request lists on BASELINE configs 1-4 and the deferred-despawn
scenarios.  The reference holds no absolute checksum / f32 vector ("parity unpinned by the
reference"); this is the second of the two pinning routes VERDICT r1 names -- two restatements by construction
different in language, storage shape and control flow must agree bit for bit before either is trusted."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, REFSHAPED, OracleWorld
from oracle.twin_np import TwinWorld

import test_box_game as tb
import test_despawn_rollback as td
from test_oracle_selfcheck import _p2p_run


def _particles_synctest(world, n, cd, ticks, rate=100, ttl_init=40, max_prediction=None):
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ids = cm.build_particles(world, with_spawn=True, ttl_init=ttl_init)
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, cd, max_prediction=max_prediction)
    fn = cm.frame_spawn_fn(rate)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
    return drv.all_checksums, cm.snapshot_state(world, ids)


def _same(a, b, ctx):
    assert len(a[0]) == len(b[0]) > 0
    for (fa, ca), (fb, cb) in zip(a[0], b[0]):
        assert fa == fb and ca == cb, f"{ctx} frame {fa}: oracle {ca:#x} twin {cb:#x}"
    cm.assert_states_equal(a[1], b[1], ctx)


@pytest.mark.parametrize("n,cd,ticks", [(1, 2, 10), (65, 1, 9), (1025, 7, 22), (10_000, 8, 26)])
def test_config2_particles_synctest(n, cd, ticks):
    """BASELINE config 2 (stress_test 10 k, SyncTest depth 8) and ragged sizes, spawns + Ttl despawns inside the
    rolled-back window."""
    cap = n + 100 * ticks + 64
    a = _particles_synctest(OracleWorld(cap, 16, FLAT), n, cd, ticks)
    b = _particles_synctest(TwinWorld(cap, 16), n, cd, ticks)
    _same(a, b, f"config2 n={n}")


def test_config3_depth8_100k():
    """BASELINE config 3's shape (3 registered components, depth 8 with max_prediction 9) at 100 k entities."""
    n = 100_000
    res = []
    for w in (OracleWorld(n, 9, FLAT), TwinWorld(n, 9)):
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        ids = cm.build_particles(w)
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, 8, max_prediction=9)
        for _ in range(11):
            drv.tick((0,))
        res.append((drv.all_checksums, cm.snapshot_state(w, ids)))
    _same(res[0], res[1], "config3")


def test_config4_p2p_shaped_rollbacks():
    """BASELINE config 4's request shape: rollbacks of 0..7 frames per tick, confirmed frame trailing by 8."""
    a, sa = _p2p_run(OracleWorld(30_000, 8, FLAT), 20_000, 30)
    b, sb = _p2p_run(TwinWorld(30_000, 8), 20_000, 30)
    assert a.depths == b.depths
    _same((a.all_checksums, sa), (b.all_checksums, sb), "config4")
    assert a.world.snapshot_count() == b.world.snapshot_count() and not b.world.has_snapshot(0)


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_config1_box_game_synctest(mode):
    """BASELINE config 1: box_game SyncTest, 2 players, check distance 7, input delay 2 (plus a many-cube variant)."""
    for n, players in ((2, 2), (3000, 4)):
        a = tb.synctest_box_game(OracleWorld(n, 16, mode), players, 7, 36, n)
        b = tb.synctest_box_game(TwinWorld(n, 16), players, 7, 36, n)
        assert a[0] == b[0]                                   # every checksum of every save
        for (ta, va), (tb_, vb) in zip(a[1], b[1]):
            assert np.array_equal(ta, tb_) and np.array_equal(va, vb)


def test_deferred_despawn_scripted_and_synctest():
    """RollbackDespawned (snapshot/despawn.rs): the scripted resurrect / confirm walk, the SyncTest Health scenario
    (tests/synctest.rs:60-75) and the particles world with host-issued markers."""
    a, b = td.scripted(OracleWorld(400, 8, FLAT)), td.scripted(TwinWorld(400, 8))
    td.check_scripted(b)
    for (na, sa), (nb, sb) in zip(a, b):
        assert na == nb
        cm.assert_states_equal(sa, sb, f"scripted {na}")
    ca, tra = td.synctest_run(OracleWorld(6000, 16, FLAT), 5000, 14, 3)
    cb, trb = td.synctest_run(TwinWorld(6000, 16), 5000, 14, 3)
    assert ca == cb
    for k, (sa, sb) in enumerate(zip(tra, trb)):
        cm.assert_states_equal(sa, sb, f"health synctest tick {k}")
    csa, outa = td.particles_markers_scripted(OracleWorld(3000, 8, FLAT))
    csb, outb = td.particles_markers_scripted(TwinWorld(3000, 8))
    td.check_markers_scripted(csb, outb)
    assert csa == csb
    for k, (sa, sb) in enumerate(zip(outa, outb)):
        cm.assert_states_equal(sa, sb, f"markers {k}")


def test_presence_masks_insert_remove_and_u64_checksum_specs():
    """component_snapshot.rs:106-115 (insert / remove across a rollback) and a checksum spec over a u64 word."""
    res = []
    for w in (OracleWorld(4000, 8, FLAT), TwinWorld(4000, 8)):
        foo = w.register_component("Foo", 4, 1); bar = w.register_component("Bar", 4, 1); ttl = w.register_component("Ttl", 8, 1)
        for c in (foo, bar): w.checksum_component(c, [0])
        w.checksum_component(ttl, [0])
        w.add_system(bg.SYS_ADD_U32, comp=(foo,), word=(0,), iparam=(1,))
        w.add_system(bg.SYS_ADD_U32, comp=(bar,), word=(0,), iparam=(-1 & 0xFFFFFFFF,))
        w.add_system(bg.SYS_TTL_DESPAWN, comp=(ttl,), word=(0,))
        v = np.arange(1000, dtype=np.uint32)
        w.spawn(1000, {foo: [v], ttl: [(2 + v % 7).astype(np.uint64) << np.uint64(31)]}); w.spawn(1000, {bar: [v], ttl: [(1 + v % 3).astype(np.uint64)]})
        w.set_depth(8)
        out = w.handle_requests([bg.SaveGameState(0), bg.AdvanceFrame((0,)), bg.SaveGameState(1)])
        w.remove_component(foo, 5); w.insert_component(bar, 5, np.array([77], np.uint32)); w.despawn(1500)
        out += w.handle_requests([bg.AdvanceFrame((0,)), bg.SaveGameState(2), bg.LoadGameState(1), bg.SaveGameState(1), bg.AdvanceFrame((0,)), bg.SaveGameState(2)])
        res.append((out, cm.snapshot_state(w, (foo, bar, ttl))))
    assert res[0][0] == res[1][0]
    assert res[0][0][1] == res[0][0][3] and res[0][0][2] != res[0][0][4]
    cm.assert_states_equal(res[0][1], res[1][1], "presence")


def test_narrow_words_one_and_two_byte_components():
    """Components whose words are 1 or 2 bytes wide (a `bool` / u8 enum, a u16: the reference's Visibility family,
    particles.rs:190-199): derive(Hash) feeds 1 / 2 bytes per field into the SeaHasher, so a checksum over them exercises the
    byte-granular stream -- restated independently in numpy (oracle_np.np_inner_hash_fields) and in C (SeaStream)."""
    res = []
    for w in (OracleWorld(3000, 8, FLAT), TwinWorld(3000, 8)):
        T, V, L = cm.build_particles(w, with_spawn=False)
        vis = w.register_component("Visibility", 1, 1); flags = w.register_component("Flags", 1, 3); tag = w.register_component("Tag", 2, 2)
        w.set_component_default(vis, np.array([1], dtype=np.uint8))
        w.checksum_component(vis, [0]); w.checksum_component(flags, [2, 0]); w.checksum_component(tag, [1, 0])
        n = 1500
        rng = np.random.default_rng(5)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        tcols = [np.full(n, cm.f32bits(cm.TRANSFORM_DEFAULT)[k], dtype=np.uint32) for k in range(10)]
        vcols = [cm.f32bits(vel[:, 0]), cm.f32bits(vel[:, 1]), np.zeros(n, dtype=np.uint32)]
        w.spawn(n, {T: tcols, V: vcols, L: [ttl], vis: None, flags: [rng.integers(0, 256, n, dtype=np.uint64).astype(np.uint8) for _ in range(3)],
                    tag: [rng.integers(0, 65536, n, dtype=np.uint64).astype(np.uint16) for _ in range(2)]})
        w.set_depth(8)
        out = w.handle_requests([bg.SaveGameState(0), bg.AdvanceFrame((0,)), bg.SaveGameState(1), bg.AdvanceFrame((0,))])
        w.insert_component(tag, 7, np.array([513, 65535], dtype=np.uint16)); w.remove_component(flags, 9)
        w.insert_component(vis, 11, np.array([0], dtype=np.uint8))
        out += w.handle_requests([bg.SaveGameState(2), bg.AdvanceFrame((0,)), bg.LoadGameState(1), bg.SaveGameState(1), bg.AdvanceFrame((0,)), bg.SaveGameState(2)])
        res.append((out, cm.snapshot_state(w, (T, V, L, vis, flags, tag))))
    assert res[0][0] == res[1][0]
    assert res[0][0][1] == res[0][0][3] and res[0][0][2] != res[0][0][4]          # frame 1 again after the rollback; frame 2 without the host edits
    cm.assert_states_equal(res[0][1], res[1][1], "narrow words")


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_allhot_schema_oracle_vs_twin(mode):
    """bench.py --schema allhot (tests/common.py): the stress_test systems plus increase_component (benches/bench.rs:30-46) over the 7 words of
    Transform.rotation / .scale -- the all-columns-hot world of DESIGN.md section 6.  Both restatements must agree on every checksum and on
    the bits of every column, spawns and Ttl despawns included, before the GPU line's parity gate means anything."""
    n, cd, ticks = 3000, 8, 20
    res = []
    for w in (OracleWorld(n + 100 * (ticks + 20), cd + 1, mode), TwinWorld(n + 100 * (ticks + 20), cd + 1)):
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        ids = cm.build_particles(w, with_spawn=True, ttl_init=40, schema="allhot")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, cd, max_prediction=cd + 1)
        fn = cm.frame_spawn_fn(100)
        for t in range(ticks):
            drv.tick((cm.INPUT_SPAWN if t % 4 == 1 else 0,), spawn_fn=fn)
        res.append((drv.all_checksums, cm.snapshot_state(w, ids)))
    _same(res[0], res[1], "allhot")
    rot_w = res[0][1]["c0w6"]                                           # rotation.w started at 1.0f and was incremented as a u32 once per net frame
    assert int(rot_w[250]) == int(cm.f32bits(cm.TRANSFORM_DEFAULT)[6]) + ticks       # (slot 250: Ttl 251, still alive)
