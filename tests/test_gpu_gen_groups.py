"""Request-group fusion for worlds OTHER than the stress_test schedule: k_tick_gen stages a workgroup's slots in LDS
and replays [Load?](Save | Advance)* there for any mix of the kernel-backed systems.  Every scenario runs on the
generic fused path (flags 0), on the one-launch-per-request path (GGRS_WORLD_NO_GROUPS) and on the oracle; all three
must agree bit for bit, and the fused worlds must really have used the group kernel."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import OracleWorld

pytestmark = pytest.mark.gpu


def _worlds(cap, depth=8):
    return [("gen", bg.World(cap, max_depth=depth)), ("per-request", bg.World(cap, max_depth=depth, flags=bg.GGRS_WORLD_NO_GROUPS)),
            ("oracle", OracleWorld(cap, depth))]


def _group_launches(w):
    ms = w.profile_read()
    return ms["tick"][1]


def _check(results):
    (n0, c0, s0), rest = results[0], results[1:]
    for name, cs, st in rest:
        assert cs == c0, f"checksums of {name} differ from {n0}"
        cm.assert_states_equal(st, s0, f"{name} vs {n0}")


@pytest.mark.parametrize("n,cd,ticks", [(1, 2, 10), (300, 3, 16), (5000, 7, 20), (70_000, 4, 12)])
def test_particles_with_extra_checksum_specs(n, cd, ticks):
    """particles + spawn, but Ttl (u64) and a non-prefix word subset of Transform are checksummed too: the register
    kernel k_tick does not cover those specs, the generic kernel does."""
    res = []
    for name, w in _worlds(n + 60 * ticks + 64):
        T = w.register_component("Transform", 4, 10); V = w.register_component("Velocity", 4, 3); L = w.register_component("Ttl", 8, 1)
        w.set_component_default(T, cm.TRANSFORM_DEFAULT)
        w.checksum_component(V, [0, 1, 2]); w.checksum_component(L, [0]); w.checksum_component(T, [2, 0, 9, 6])
        w.add_system(bg.SYS_PARTICLES_UPDATE, comp=(T, V), word=(0, 0), fparam=(0.0, -200.0, 0.0))
        w.add_system(bg.SYS_TTL_DESPAWN, comp=(L,), word=(0,))
        w.add_system(bg.SYS_PARTICLES_SPAWN, comp=(T, V, L), iparam=(30, cm.INPUT_SPAWN))
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, (T, V, L), n, vel, ttl)
        if name != "oracle": w.profile_enable(True)
        drv = cm.SyncTestDriver(w, cd)
        fn = cm.frame_spawn_fn(60)
        for t in range(ticks):
            drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
        if name == "gen": assert _group_launches(w) > 0
        if name == "per-request": assert _group_launches(w) == 0
        res.append((name, drv.all_checksums, cm.snapshot_state(w, (T, V, L))))
    _check(res)


@pytest.mark.parametrize("n,cd", [(200, 3), (5000, 2), (70_000, 7)])
def test_health_world_immediate_despawn(n, cd):
    """tests/synctest.rs:26-75: Health, decrease_health with commands.entity(e).despawn() at zero."""
    res = []
    for name, w in _worlds(n + 8):
        H = w.register_component("Health", 4, 1)
        w.checksum_component(H, [0])
        w.add_system(bg.SYS_SAT_SUB_DESPAWN, comp=(H,), word=(0,), iparam=(1, bg.DESPAWN_IMMEDIATE))
        w.spawn(n, {H: [(1 + (np.arange(n) % 9)).astype(np.uint32)]})
        if name != "oracle": w.profile_enable(True)
        drv = cm.SyncTestDriver(w, cd)
        for _ in range(16):
            drv.tick((0,))
        if name == "gen": assert _group_launches(w) > 0
        st = cm.snapshot_state(w, (H,))
        assert not st["alive"].any()                       # everybody died by frame 9
        res.append((name, drv.all_checksums, st))
    _check(res)


def test_disjoint_components_in_request_lists():
    """benches/bench.rs:68-95 foo_bar_baz (three add_u32 systems over disjoint entity sets) driven by request lists;
    8-byte and 4-byte words side by side, one component never touched by any system."""
    res = []
    for name, w in _worlds(4000):
        foo = w.register_component("Foo", 4, 1); bar = w.register_component("Bar", 4, 2); baz = w.register_component("Baz", 4, 1)
        big = w.register_component("Big", 8, 2)
        for c in (foo, bar, baz): w.checksum_component(c, list(range(1 if c != bar else 2)))
        w.checksum_component(big, [1])
        w.add_system(bg.SYS_ADD_U32, comp=(foo,), word=(0,), iparam=(1,))
        w.add_system(bg.SYS_ADD_U32, comp=(bar,), word=(1,), iparam=(-1 & 0xFFFFFFFF,))
        w.add_system(bg.SYS_ADD_U32, comp=(baz,), word=(0,), iparam=(3,))
        v = np.arange(1000, dtype=np.uint32)
        w.spawn(1000, {foo: [v], big: [v.astype(np.uint64) << np.uint64(33), v.astype(np.uint64) * np.uint64(7)]})
        w.spawn(1000, {bar: [v, v + 5]})
        w.spawn(1500, {baz: [np.arange(1500, dtype=np.uint32)], foo: [np.arange(1500, dtype=np.uint32) * 2]})
        if name != "oracle": w.profile_enable(True)
        drv = cm.SyncTestDriver(w, 5)
        for _ in range(18):
            drv.tick((0,))
        if name == "gen": assert _group_launches(w) > 0
        res.append((name, drv.all_checksums, cm.snapshot_state(w, (foo, bar, baz, big))))
    _check(res)


def test_box_game_uses_the_group_kernel():
    from test_box_game import build_box, input_script
    res = []
    for name, w in _worlds(600):
        ids, *_ = build_box(w, 500, 4, spread=True)
        if name != "oracle": w.profile_enable(True)
        drv = cm.SyncTestDriver(w, 6, num_players=4, input_delay=2)
        for t in range(24):
            drv.tick(input_script(t, 4))
        if name == "gen": assert _group_launches(w) > 0
        res.append((name, drv.all_checksums, cm.snapshot_state(w, ids[:2])))
    _check(res)


def test_long_request_lists_split_into_groups():
    """One handle_requests call with far more ops than a group holds (16 saves / 24 steps / 40 ops) and a
    LoadGameState in the middle: the list is cut into several launches; results must not depend on the cuts."""
    res = []
    for name, w in _worlds(3000, depth=60):
        foo = w.register_component("Foo", 4, 2); big = w.register_component("Big", 8, 1)
        w.checksum_component(foo, [0, 1]); w.checksum_component(big, [0])
        w.add_system(bg.SYS_ADD_U32, comp=(foo,), word=(1,), iparam=(7,))
        w.add_system(bg.SYS_TTL_DESPAWN, comp=(big,), word=(0,))
        n = 2500
        w.spawn(n, {foo: [np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32) * 3],
                    big: [(5 + np.arange(n, dtype=np.uint64) % 90)]})
        w.set_depth(60)
        w.set_confirmed(0)
        reqs = []
        for f in range(45):
            reqs += [bg.SaveGameState(f), bg.AdvanceFrame((0,))]
        reqs += [bg.LoadGameState(20)]
        for f in range(20, 70):
            reqs += [bg.AdvanceFrame((0,)), bg.SaveGameState(f + 1)]
        reqs += [bg.AdvanceFrame((0,))] * 30                      # more steps than one group takes, no saves
        reqs += [bg.SaveGameState(100)]
        cs = w.handle_requests(reqs)
        assert len(cs) == 45 + 50 + 1
        res.append((name, cs, cm.snapshot_state(w, (foo, big))))
    _check(res)
    assert res[0][2]["frame"] == 100 and res[0][1][21] == res[0][1][45]      # frame 21: saved before the load and again after it


@pytest.mark.parametrize("n", [450_000, 1_100_000])
def test_hbm_sized_generic_world_on_the_generated_kernel(n):
    """A world beyond the stress_test schema (an extra system, an extra checksum spec, a fourth component) at HBM size: the default
    dispatch runs the kernel generated for it (one slot per lane, non-temporal snapshot stores above 416 k slots, no
    depth-parallel roles at this size) -- against the oracle, depth-8 SyncTest with despawns."""
    cd, ticks = 8, 11
    res = []
    for name, w in [("gen", bg.World(n, max_depth=9)), ("oracle", OracleWorld(n, 9))]:
        T = w.register_component("Transform", 4, 10); V = w.register_component("Velocity", 4, 3); L = w.register_component("Ttl", 8, 1)
        H = w.register_component("Health", 4, 1)
        w.set_component_default(T, cm.TRANSFORM_DEFAULT)
        w.checksum_component(V, [0, 1, 2]); w.checksum_component(T, [0, 1, 2]); w.checksum_component(H, [0]); w.checksum_component(L, [0])
        w.add_system(bg.SYS_PARTICLES_UPDATE, comp=(T, V), word=(0, 0), fparam=(0.0, -200.0, 0.0))
        w.add_system(bg.SYS_TTL_DESPAWN, comp=(L,), word=(0,))
        w.add_system(bg.SYS_ADD_U32, comp=(H,), word=(0,), iparam=(3,))
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        tcols = [np.full(n, cm.f32bits(cm.TRANSFORM_DEFAULT)[k], dtype=np.uint32) for k in range(10)]
        vcols = [cm.f32bits(vel[:, 0]), cm.f32bits(vel[:, 1]), np.zeros(n, dtype=np.uint32)]
        w.spawn(n, {T: tcols, V: vcols, L: [ttl], H: [np.arange(n, dtype=np.uint32)]})
        if name != "oracle": w.profile_enable(True)
        drv = cm.SyncTestDriver(w, cd)
        for t in range(ticks):
            drv.tick((0,))
        if name == "gen": assert _group_launches(w) >= ticks
        res.append((name, drv.all_checksums, cm.snapshot_state(w, (T, V, L, H))))
    _check(res)


@pytest.mark.parametrize("sync", [True, False])
def test_steady_group_shape_gets_its_own_kernel(sync, monkeypatch):
    """include/ggrs_hip.h ggrs_hip_specialise_wait: after GGRS_JIT_SPECIALISE_AFTER identical HBM-sized groups the library builds the
    generated kernel again with that shape's op sequence and row masks as literals -- on the calling thread (sync) or on a worker
    (the default) -- and runs the shape on it from then on.  Depth-8 SyncTest with Ttl despawns against the oracle across the switch,
    then another shape (check distance 5: the old kernel must not be used for it, a new one replaces it), then a spawn that changes
    the row masks for a few ticks (general kernel) and the steady state again."""
    monkeypatch.setenv("GGRS_JIT_SPECIALISE_AFTER", "3")
    monkeypatch.setenv("GGRS_JIT_SPECIALISE_SYNC", "1" if sync else "0")
    n = 600_000
    res, infos = [], []
    for name, w in [("gen", bg.World(n + 4000, max_depth=9)), ("oracle", OracleWorld(n + 4000, 9))]:
        ids = cm.build_particles(w, with_spawn=True, ttl_init=50)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        fn = cm.frame_spawn_fn(64)
        drv = cm.SyncTestDriver(w, 8)
        for t in range(9): drv.tick((0,))
        if name == "gen":
            ready = w.specialise_wait()
            infos.append((ready, w.kernel_info()["specialised_kernel"]))
        for t in range(6): drv.tick((0,))
        out = list(drv.all_checksums)
        # another shape, driven by hand: roll back 5 frames every tick ([Load(F-5), (Adv, Save) x 5, Adv])
        if hasattr(w, "set_synctest_check_distance"): w.set_synctest_check_distance(-1)

        def roll5(spawn=False):
            F = w.frame
            w.set_confirmed(max(0, F - 7))
            reqs = [bg.LoadGameState(F - 5)]
            for i in range(5):
                a = bg.AdvanceFrame((cm.INPUT_SPAWN if (spawn and i == 2) else 0,))
                if spawn and i == 2: a.spawn_vx, a.spawn_vy = fn(F - 5 + i)
                reqs += [a, bg.SaveGameState(F - 4 + i)]
            reqs.append(bg.AdvanceFrame((0,)))
            return w.handle_requests(reqs)

        for t in range(9): out += roll5()
        if name == "gen": infos.append((w.specialise_wait(), w.kernel_info()["specialised_kernel"]))
        for t in range(4): out += roll5()
        out += roll5(spawn=True)                                               # new rows: every column of the bundle is stored again for a while
        for t in range(8): out += roll5()
        if name == "gen": infos.append((w.specialise_wait(), w.kernel_info()["specialised_kernel"]))
        res.append((name, out, cm.snapshot_state(w, ids)))
    _check(res)
    assert all(r for r, _ in infos) and all(s.startswith("ready") for _, s in infos), infos


@pytest.mark.parametrize("n,table", [(100_000, 16), (450_000, 16), (100_000, 3)])
def test_every_rollback_length_of_a_p2p_session_gets_its_own_kernel(n, table, monkeypatch):
    """host_world.hpp JitSpecSlot: shapes are counted one by one, so a P2P session -- whose rollback length changes from tick to tick
    (BASELINE config 4: [Load(F-r), Adv, (Save, Adv) x (r-1), Save(F), Adv], r drawn per tick) -- ends up with one specialised kernel per
    length instead of none.  Ttl despawns all along, a spawn in the middle (its row masks are shapes of their own for a few ticks);
    every Save's checksum and the final state against the oracle, across all the kernel switches.  table = 3: eight lengths fight for
    three places -- kernels are unloaded (after the stream has drained) and built again tick after tick until the per-world build
    budget (host_world.hpp JIT_SPEC_MAX_BUILDS) is spent."""
    monkeypatch.setenv("GGRS_JIT_SPECIALISE_AFTER", "2")
    monkeypatch.setenv("GGRS_JIT_SPECIALISE_SYNC", "1")
    res, info = [], None
    for name, w in [("gen", bg.World(n + 4000, max_depth=9)), ("oracle", OracleWorld(n + 4000, 9))]:
        if name == "gen": assert w._lib.ggrs_dbg_set_spec_shapes(w._p, table) == 0       # test hook: places in the shape table (default 16)
        ids = cm.build_particles(w, with_spawn=True, ttl_init=50)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.P2PShapeDriver(w, 8, inputs=lambda f: (cm.INPUT_SPAWN if f in (60, 61) else 0, 0), spawn_fn=cm.frame_spawn_fn(64))
        for _ in range(110): drv.tick()
        if name == "gen": info = w.kernel_info()["specialised_kernel"]
        res.append((name, drv.all_checksums, cm.snapshot_state(w, ids)))
    _check(res)
    assert info.startswith("ready (") and int(info[len("ready ("):].split()[0]) >= (6 if table == 16 else 1), info
    assert int(info.split(" of ")[1].split()[0]) <= table, info
