"""Row versions (csrc/host_world.hpp): a SaveWorld / LoadWorld moves only the columns whose bytes differ between source and
destination.  Every way a live column can change behind a fused group's back -- uploads, inserts, spawns with and without the
component in the bundle, a handed-out device pointer, adopt_live_state, ring depth changes, rollbacks into and out of all of it
-- driven through the default library, the library with GGRS_ROW_VERSIONS=0 and the CPU oracle: every Checksum(u128), every
column of the live world AND of every frame the ring still holds must agree bit for bit."""
import ctypes as C

import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu


def _ring_contents(w, ids, frames):
    """Load every frame the ring holds (newest first: a rollback pops what is newer) and record the whole state."""
    out = {}
    for f in sorted(frames, reverse=True):
        if not w.has_snapshot(f): continue
        w.handle_requests([bg.LoadGameState(f)])
        st = cm.snapshot_state(w, ids)
        out[f] = {k: (v.tobytes() if hasattr(v, "tobytes") else v) for k, v in st.items()}
    return out


def _script(w, n, flags_desc):
    """Particles + an `Extra` component no system writes + a checksummed `Tag`; returns (checksums, live state, ring contents)."""
    rng = np.random.default_rng(21)
    T, V, L = cm.build_particles(w, with_spawn=True, ttl_init=30)
    X = w.register_component("Extra", 4, 3)
    G = w.register_component("Tag", 2, 1)
    w.checksum_component(X, [2, 0])
    w.checksum_component(G, [0])
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    tcols = [np.full(n, cm.f32bits(cm.TRANSFORM_DEFAULT)[k], dtype=np.uint32) for k in range(10)]
    vcols = [cm.f32bits(vel[:, 0]), cm.f32bits(vel[:, 1]), np.zeros(n, dtype=np.uint32)]
    xcols = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for _ in range(3)]
    w.spawn(n, {T: tcols, V: vcols, L: [ttl], X: xcols, G: [rng.integers(0, 65536, n, dtype=np.uint64).astype(np.uint16)]})
    ids = (T, V, L, X, G)
    w.set_depth(6); w.set_confirmed(0)
    fn = cm.frame_spawn_fn(40)
    out = []

    def adv(spawn=False):
        a = bg.AdvanceFrame((cm.INPUT_SPAWN if spawn else 0,))
        if spawn: a.spawn_vx, a.spawn_vy = fn(w.frame)
        return a

    def tick(k=1, spawn=False):
        nonlocal out
        for _ in range(k):
            out += w.handle_requests([bg.SaveGameState(w.frame), adv(spawn)])

    tick(4)                                                        # frames 0..3 saved: Extra / Tag reach four ring slots
    w.upload_word(X, 1, 0, np.arange(n, dtype=np.uint32))          # host edit: the NEXT save must carry it, frames 0..3 must not
    tick(2)                                                        # 4, 5
    out += w.handle_requests([bg.LoadGameState(2), adv(), bg.SaveGameState(3), adv(), bg.SaveGameState(4), adv()])   # back before the edit: it is gone again
    w.insert_component(X, 5, np.array([7, 8, 9], dtype=np.uint32)); w.remove_component(G, 9)
    tick(2, spawn=True)                                            # 5, 6: the spawn system appends rows WITHOUT Extra / Tag
    w.spawn(17, {T: None, V: None, L: [np.full(17, 1 << 20, dtype=np.uint64)], X: None})      # API spawn: Extra in the bundle (defaults), Tag not
    tick(1)
    # a device pointer to a column is handed out and written behind the library's back
    if isinstance(w, bg.World):
        ptr, stride = w.column_device_ptr(X, 0)
        host = np.full(min(n, 8192), 0xDEADBEEF, dtype=np.uint32)
        hip = C.CDLL(None)
        assert hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), C.c_int(1)) == 0   # H2D into layout tile 0
    else:
        w.upload_word(X, 0, 0, np.full(min(n, 8192), 0xDEADBEEF, dtype=np.uint32))
    tick(2)
    w.set_depth(3); tick(3); w.set_depth(6); tick(2)               # the ring shrinks and grows: slots are recycled with stale versions
    out += w.handle_requests([bg.LoadGameState(w.frame - 2), adv(), bg.SaveGameState(w.frame + 1), adv()])
    # presence masks carry versions of their own (only the host changes them): a removal and an insertion with NO column write in
    # between saves, a rollback to before both (the masks come back), and fused groups saving into slots that still hold the old masks
    w.remove_component(X, 11); w.remove_component(T, 3)
    tick(2)
    w.insert_component(G, 9, np.array([4242], dtype=np.uint16))
    tick(1)
    out += w.handle_requests([bg.LoadGameState(w.frame - 4), adv(), bg.SaveGameState(w.frame + 1), adv(), bg.SaveGameState(w.frame + 2), adv()])
    w.remove_component(V, 2)
    out += w.handle_requests([bg.SaveGameState(w.frame), adv(), bg.SaveGameState(w.frame + 1), adv(), bg.LoadGameState(w.frame + 1), adv(), bg.SaveGameState(w.frame + 2), adv()])
    if isinstance(w, bg.World):                                    # "an external producer rewrote the live block": nothing may be assumed
        w.live_state_ptr()                                         # (refreshes the block's header -- here the producer writes back what was there)
        w.adopt_live_state()
    tick(2)
    live = cm.snapshot_state(w, ids)
    frames = list(range(0, w.frame + 1))
    ring = _ring_contents(w, ids, frames)
    return out, live, ring


@pytest.mark.parametrize("n,flags", [(3000, 0), (70_000, 0), (500_000, 0), (3000, bg.GGRS_WORLD_NO_GROUPS)])
def test_row_versions_agree_with_copy_everything_and_the_oracle(n, flags, monkeypatch):
    cap = n + 2000
    a = _script(bg.World(cap, max_depth=7, flags=flags), n, "versions on")
    monkeypatch.setenv("GGRS_ROW_VERSIONS", "0")
    b = _script(bg.World(cap, max_depth=7, flags=flags), n, "versions off")
    monkeypatch.delenv("GGRS_ROW_VERSIONS")
    o = _script(OracleWorld(cap, 7, FLAT), n, "oracle")
    assert a[0] == o[0] and b[0] == o[0]
    cm.assert_states_equal(a[1], o[1], "live, versions on"); cm.assert_states_equal(b[1], o[1], "live, versions off")
    assert sorted(a[2]) == sorted(o[2]) == sorted(b[2]) and len(o[2]) >= 3
    for f in o[2]:
        assert a[2][f] == o[2][f], f"ring frame {f} (row versions on)"
        assert b[2][f] == o[2][f], f"ring frame {f} (row versions off)"


def _tagged(cap, depth, flags=0):
    """A world with VALUE TAGS forced on whatever its size (test hook; by default only worlds whose steady Save is bound by bytes keep them)."""
    w = bg.World(cap, max_depth=depth, flags=flags)
    assert w._lib.ggrs_dbg_set_value_tags(w._p, 1) == 0
    return w


@pytest.mark.parametrize("n", [3000, 70_000, 500_000])
def test_value_tags_agree_with_the_oracle_on_the_row_version_script(n):
    """Value tags forced on: a Save skips the columns whose 64 values per unit the destination slot already holds.  The script edits the world between and
    inside request lists by every route there is -- uploads, API spawns, inserts / removes, a column written through a handed-out pointer, an adopted live
    block, a ring that shrinks and grows -- and every route that is not the generated kernel must take the tags of what it wrote with it: checksums, live
    state and EVERY ring frame's bytes equal the oracle's."""
    cap = n + 2000
    a = _script(_tagged(cap, 7), n, "value tags")
    o = _script(OracleWorld(cap, 7, FLAT), n, "oracle")
    assert a[0] == o[0]
    cm.assert_states_equal(a[1], o[1], "live, value tags")
    assert sorted(a[2]) == sorted(o[2]) and len(o[2]) >= 3
    for f in o[2]:
        assert a[2][f] == o[2][f], f"ring frame {f} (value tags on)"


def test_value_tags_skip_the_columns_whose_values_never_change():
    """The stress_test is a 2-D simulation: translation.z, velocity.x and velocity.z are in the systems' write sets and never change.  With value tags a steady
    tick stores 20 of the 32 hot bytes per entity and Save (ggrs_hip_profile_read_bytes counts what the launches really stored), every snapshot still holds
    all of its bytes (the ring frames are compared with the oracle's), and an upload into one of the constant columns is stored again exactly once per slot."""
    n, D = 70_000, 8
    res = []
    for w in (_tagged(n, D + 1), OracleWorld(n, D + 1, FLAT)):
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
        for _ in range(2 * D + 4): drv.tick((0,))
        if isinstance(w, bg.World):
            assert w.kernel_info()["value_tags"].startswith("on"), w.kernel_info()["value_tags"]
            w.profile_enable(True)
            for _ in range(5): drv.tick((0,))
            steady = w.profile_bytes()["tick"]
            w.profile_enable(False)
            # 32 B loaded + D Saves x (32 - 12) stored + the live block's 32 - 12 ... the lazy live block is off at this size, so: 32 + (D + 1) x 20
            assert steady == 5 * n * (32 + (D + 1) * 20), (steady, 5 * n * (32 + (D + 1) * 20))
        else:
            for _ in range(5): drv.tick((0,))
        w.upload_word(ids[1], 2, 100, np.full(5000, 0x40400000, dtype=np.uint32))     # velocity.z of 5000 entities: the column is no longer what the slots hold
        for _ in range(D + 3): drv.tick((0,))
        res.append((list(drv.all_checksums), cm.snapshot_state(w, ids), _ring_contents(w, ids, list(range(w.frame - D, w.frame)))))
    assert res[0][0] == res[1][0]
    cm.assert_states_equal(res[0][1], res[1][1], "live")
    assert sorted(res[0][2]) == sorted(res[1][2]) and len(res[1][2]) >= D - 1
    for f in res[1][2]: assert res[0][2][f] == res[1][2][f], f"ring frame {f}"


@pytest.mark.parametrize("n", [3000, 300_000])
def test_value_tag_ids_start_over_without_leaving_a_tag_behind(n):
    """The ids are 32-bit and count up per launch: when they start over, every block's tags are zeroed on the stream and its tag_ok bits cleared (host_groups.hpp
    vtags_reserve) -- a tag of the old numbering left in a unit no launch covers for a while must not meet the same number again.  Every world crosses its FIRST
    start-over within its first few dozen launches (host_seal.hpp), so the path runs here (and in every other test of a tag-keeping world): a SyncTest session with an
    upload into a constant column on either side of it, checksums / live state / every ring frame equal to the oracle's."""
    D = 8
    res = []
    for w in (_tagged(n, D + 1), OracleWorld(n, D + 1, FLAT)):
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
        for k in range(70):
            if k in (20, 45): w.upload_word(ids[1], 2, 64 * (k % 7), np.full(min(n, 1000), 0x40400000 + k, dtype=np.uint32))
            drv.tick((0,))
        if isinstance(w, bg.World):
            info = w.kernel_info()["value_tags"]
            assert info.startswith("on") and "has started over 1 times" in info, info
        res.append((list(drv.all_checksums), cm.snapshot_state(w, ids), _ring_contents(w, ids, list(range(w.frame - D, w.frame)))))
    assert res[0][0] == res[1][0]
    cm.assert_states_equal(res[0][1], res[1][1], "live")
    assert sorted(res[0][2]) == sorted(res[1][2]) and len(res[1][2]) >= D - 1
    for f in res[1][2]: assert res[0][2][f] == res[1][2][f], f"ring frame {f}"


def test_steady_state_tick_moves_only_what_systems_write():
    """The point of it: in a steady SyncTest tick of the stress_test the library asks its kernel for 320 B per entity (32 loaded +
    8 x 32 + 32 stored), not 600 -- as counted by ggrs_hip_profile_read_bytes, the numerator of bench.py's roofline."""
    n = 20_000
    w = bg.World(n, max_depth=9)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, 8, max_prediction=9)
    for _ in range(12): drv.tick((0,))
    w.profile_enable(True)
    for _ in range(5): drv.tick((0,))
    prof, nbytes = w.profile_read(), w.profile_bytes()
    w.profile_enable(False)
    assert prof["tick"][1] == 5 and nbytes["tick"] == 5 * 320 * n, (prof, nbytes)
