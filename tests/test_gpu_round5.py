"""Round-5 surface of the drop-in boundary (ABI v8), every case against the CPU oracle bit for bit:

* a user-written SPAWN system with an opaque payload next to a user-written despawn system -- `commands.spawn((.., Rollback))` /
  `commands.entity(e).despawn()` from any GgrsSchedule system (/root/reference/src/snapshot/rollback.rs:45-59,
  examples/stress_tests/particles.rs:254-289) -- fused into ONE launch per tick;
* PlayerInputs<T> as the reference defines it (src/lib.rs:98): inputs of more than one byte AND the InputStatus of every player reach a
  user-written system (schedule_systems.rs:251-268);
* a Strategy whose Stored differs from its Target (src/snapshot/strategy.rs:22-40): an f32 x 3 component snapshotted as 3 x f16;
* the host timeline and the shipped-code-object path (no run-time compiler)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu

f32 = np.float32


def bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def unbits(b):
    return np.array([b & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]


# ------------------------------------------------------------------------------------------------ bullets
FIRE = 1 << 4
BULLET_MOVE = """
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {           // Query<(&mut Pos, &Vel, &mut Life)>
    e.f32(0) = e.f32(0) + e.f32(2) * f.dt;                                 // pos += vel * dt (two roundings: the build never contracts)
    e.f32(1) = e.f32(1) + e.f32(3) * f.dt;
    if (e.u32(4) <= 1u) e.despawn(); else e.u32(4) -= 1u;                  // commands.entity(e).despawn()
}
"""
BULLET_SPAWN = """
struct Shot { float x, y, vx, vy; };
__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload) {      // commands.spawn((Pos, Vel, Life, Rollback))
    const Shot* s = reinterpret_cast<const Shot*>(payload);                // this entity's record (payload_stride = 16)
    e.f32(0) = s->x; e.f32(1) = s->y; e.f32(2) = s->vx; e.f32(3) = s->vy;
    e.u32(4) = (ggrs_u32)f.iparam[0] + (ggrs_u32)(k & 3u) + (ggrs_u32)(f.frame & 1);
}
"""


def _bullet_world(w, life):
    P = w.register_component("Pos", 4, 2)
    V = w.register_component("Vel", 4, 2)
    L = w.register_component("Life", 4, 1)
    K = w.register_component("Kind", 1, 1)                                # not part of the bullets' bundle: spawned entities must come out WITHOUT it
    w.set_component_default(L, np.array([7], dtype=np.uint32))
    w.checksum_component(P, [0, 1]); w.checksum_component(L, [0]); w.checksum_component(K, [0])
    binds = [(P, 0), (P, 1), (V, 0), (V, 1), (L, 0)]
    if isinstance(w, OracleWorld):
        def move(words, slot, f):
            dt = f32(f.dt)
            x = f32(unbits(words[0]) + f32(unbits(words[2]) * dt)); y = f32(unbits(words[1]) + f32(unbits(words[3]) * dt))
            life_ = words[4] & 0xFFFFFFFF
            kill = 1 if life_ <= 1 else 0
            return [bits(x), bits(y), words[2], words[3], life_ if kill else life_ - 1], kill

        def spawn(words, slot, k, f, payload):
            rec = bytes(payload[:16]); x, y, vx, vy = struct.unpack("<4I", rec)
            return [x, y, vx, vy, (int(f.iparam[0]) + (k & 3) + (f.frame & 1)) & 0xFFFFFFFF]
        w.add_custom_system(move, binds)
        w.add_spawn_system(spawn, bundle=(P, V, L), bindings=binds, payload_stride=16, iparam=(life,))
    else:
        w.add_custom_system(BULLET_MOVE, binds, name="move_bullets")
        w.add_spawn_system(BULLET_SPAWN, bundle=(P, V, L), bindings=binds, payload_stride=16, iparam=(life,), name="fire")
    return P, V, L, K


def _shots(frame, inputs):
    """The host side of the spawn system: a PURE function of the frame and its inputs (a resimulated frame fires the same shots)."""
    n = sum(3 for i in inputs if i & FIRE)                                 # three bullets per firing player
    r = np.random.default_rng([7, frame])
    return n, r.uniform(-50, 50, (n, 4)).astype(np.float32)


@pytest.mark.parametrize("n,D", [(2000, 4), (70_000, 8)])
def test_bullets_user_written_spawn_and_despawn_in_one_launch_per_tick(n, D):
    ticks = 30
    res = []
    for w in (bg.World(n + 6 * (ticks + D + 2) * 2, max_depth=D + 1), OracleWorld(n + 6 * (ticks + D + 2) * 2, D + 1, FLAT)):
        P, V, L, K = _bullet_world(w, life=9)
        rng = np.random.default_rng(3)
        pos = rng.uniform(-10, 10, (n, 2)).astype(np.float32); vel = rng.uniform(-5, 5, (n, 2)).astype(np.float32)
        life = (2 + np.arange(n) % 23).astype(np.uint32)
        w.spawn(n, {P: [cm.f32bits(pos[:, 0]), cm.f32bits(pos[:, 1])], V: [cm.f32bits(vel[:, 0]), cm.f32bits(vel[:, 1])], L: [life], K: [(np.arange(n) % 5).astype(np.uint8)]})
        drv = cm.SyncTestDriver(w, D, num_players=2, max_prediction=D + 1)

        def patch(frame, r):
            cnt, shots = _shots(frame, r.inputs)
            if cnt: r.spawn_count, r.spawn_payload = cnt, shots
        if isinstance(w, bg.World): w.profile_enable(True)
        for t in range(ticks):
            drv.tick((FIRE if t % 3 == 0 else 0, FIRE if t % 5 == 1 else 0), patch=patch)
        if isinstance(w, bg.World):
            launches = w.profile_read(); w.profile_enable(False)
            info = w.kernel_info()
            assert info["spawn_system"].startswith("runs inside"), info
            steady = ticks - D - 1
            assert launches["tick"][1] <= ticks + 2 and launches["tick"][1] >= steady, launches         # ONE fused launch per tick: spawns and despawns included
            assert launches["advance"][1] == 0 and launches["save"][1] == 0 and launches["load"][1] == 0, launches
        res.append((drv.all_checksums, cm.snapshot_state(w, (P, V, L, K))))
    assert len(res[0][0]) == len(res[1][0]) > ticks
    for (fa, ca), (fb, cb) in zip(*[r[0] for r in res]):
        assert fa == fb and ca == cb, f"frame {fa}: gpu {ca:#x} oracle {cb:#x}"
    cm.assert_states_equal(res[0][1], res[1][1], "bullets")
    assert res[0][1]["len"] > n                                                   # bullets were spawned ...
    assert not res[0][1]["present3"][n:].any() and res[0][1]["present3"][:n][res[0][1]["alive"][:n]].all()     # ... without the component their bundle does not carry


def test_spawn_payload_validation():
    w = bg.World(1000, max_depth=4)
    _bullet_world(w, life=5)
    w.spawn(10, {0: None, 1: None, 2: None})
    with pytest.raises(bg.GgrsHipError) as e:
        w.handle_requests([bg.AdvanceFrame((FIRE, 0), spawn_count=3)])                          # the spawner reads 3 x 16 payload bytes: none given
    assert e.value.code == bg.GGRS_E_INVALID and "payload" in str(e.value)
    with pytest.raises(bg.GgrsHipError) as e:
        w.handle_requests([bg.AdvanceFrame((FIRE, 0), spawn_count=5000, spawn_payload=np.zeros((5000, 4), dtype=np.float32))])
    assert e.value.code == bg.GGRS_E_CAPACITY
    assert w.len == 10 and w.frame == 0                                                        # neither list touched the world
    w2 = bg.World(100, max_depth=4)
    A = w2.register_component("A", 4, 1)
    w2.add_spawn_system("__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame&, const unsigned char*) { e.u32(0) = (ggrs_u32)k; }", bundle=(A,), bindings=[(A, 0)])
    with pytest.raises(bg.GgrsHipError):                                                         # one spawn system per world
        w2.add_system(bg.SYS_PARTICLES_SPAWN, comp=(A, A, A), iparam=(1, 1))


# ------------------------------------------------------------------------------------------------ PlayerInputs<T>
INPUT_SYS = """
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {           // Query<(&mut Score, &Player)>: PlayerInputs<T>[player.handle]
    const int h = (int)e.u64(1);
    if (h >= (int)f.n_inputs) return;
    const ggrs_u32 in = f.input_u32(h);
    const int st = f.input_status(h);
    if (st == GGRS_INPUT_DISCONNECTED) e.u32(0) ^= 0xDEADu;
    else if (st == GGRS_INPUT_PREDICTED) e.u32(0) += in & 0xFFFFu;
    else e.u32(0) += (in >> 16) + f.input[h];                              // f.input[h]: the input's first byte
}
"""


def test_four_byte_inputs_and_input_status_reach_a_user_system():
    n, D, ticks = 5000, 5, 24
    res = []
    for w in (bg.World(n, max_depth=D + 1), OracleWorld(n, D + 1, FLAT)):
        w.set_input_layout(4, 3)
        S = w.register_component("Score", 4, 1)
        H = w.register_component("Player", 8, 1)
        w.checksum_component(S, [0])
        if isinstance(w, OracleWorld):
            def sysfn(words, slot, f):
                h = words[1]
                if h >= f.n_inputs: return words, 0
                inp = int.from_bytes(f.input(h), "little"); st = f.status[h]
                s = words[0] & 0xFFFFFFFF
                if st == bg.INPUT_DISCONNECTED: s ^= 0xDEAD
                elif st == bg.INPUT_PREDICTED: s = (s + (inp & 0xFFFF)) & 0xFFFFFFFF
                else: s = (s + (inp >> 16) + (inp & 0xFF)) & 0xFFFFFFFF
                return [s, h], 0
            w.add_custom_system(sysfn, [(S, 0), (H, 0)])
        else:
            w.add_custom_system(INPUT_SYS, [(S, 0), (H, 0)], name="score")
        w.spawn(n, {S: [np.arange(n, dtype=np.uint32)], H: [(np.arange(n) % 4).astype(np.uint64)]})      # handle 3 has no input: skipped
        drv = cm.SyncTestDriver(w, D, num_players=3, max_prediction=D + 1)

        def patch(frame, r):
            rr = np.random.default_rng([11, frame])
            r.inputs = tuple(int(x).to_bytes(4, "little") for x in rr.integers(0, 1 << 32, 3, dtype=np.uint64))
            r.status = tuple(int(x) for x in rr.integers(0, 3, 3))
        for t in range(ticks): drv.tick((0, 0, 0), patch=patch)
        res.append((drv.all_checksums, cm.snapshot_state(w, (S, H))))
    assert res[0][0] == res[1][0] and len(res[0][0]) > ticks
    cm.assert_states_equal(res[0][1], res[1][1], "inputs")
    with pytest.raises(bg.GgrsHipError):                                                       # an InputStatus byte outside the enum is refused before anything runs
        w0 = bg.World(10, max_depth=2); w0.register_component("X", 4, 1); w0.spawn(1, {0: None})
        w0.handle_requests([bg.AdvanceFrame((1,), status=(7,))])


# ------------------------------------------------------------------------------------------------ Strategy
F16_STRATEGY = """
__device__ unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short b; __builtin_memcpy(&b, &h, 2); return b; }
__device__ float h2f(unsigned short b) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }
__device__ void ggrs_store(const GgrsWords& t, GgrsWords& s) { for (int k = 0; k < 3; ++k) s.u16(k) = f2h(t.f32(k)); }        // Strategy::store
__device__ void ggrs_load(const GgrsWords& s, GgrsWords& t) { for (int k = 0; k < 3; ++k) t.f32(k) = h2f(s.u16(k)); }         // Strategy::load
"""


def _f16_world(w, strategy):
    A = w.register_component("Accel", 4, 3)                                 # f32 x 3, snapshotted as 3 x f16
    C_ = w.register_component("Count", 4, 1)
    w.checksum_component(A, [0, 1, 2]); w.checksum_component(C_, [0])
    if strategy:
        if isinstance(w, OracleWorld):
            st = lambda tg: [int(np.array([unbits(x)], dtype=np.float32).astype(np.float16).view(np.uint16)[0]) for x in tg]
            ld = lambda sv: [bits(np.array([x], dtype=np.uint16).view(np.float16).astype(np.float32)[0]) for x in sv]
            w.register_component_strategy(A, 2, 3, st, ld)
        else:
            w.register_component_strategy(A, 2, 3, F16_STRATEGY)
    w.add_system(bg.SYS_ADD_U32, comp=(C_,), word=(0,), iparam=(1,))
    if isinstance(w, OracleWorld):
        w.add_custom_system(lambda words, slot, f: ([bits(f32(unbits(words[0]) + f32(0.5))), bits(f32(unbits(words[1]) * f32(1.0009765625)))], 0), [(A, 0), (A, 1)])
    else:
        w.add_custom_system("__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) { e.f32(0) = e.f32(0) + 0.5f; e.f32(1) = e.f32(1) * 1.0009765625f; }", [(A, 0), (A, 1)], name="drift")
    return A, C_


@pytest.mark.parametrize("n", [3000, 100_000])
def test_strategy_f32x3_snapshotted_as_f16x3(n):
    """Accel.x holds multiples of 0.5 (exact in f16: store / load are a bijection there -> the session behaves as without the strategy);
    Accel.y drifts by a factor that is NOT representable in f16 (a lossy snapshot: every LoadWorld rounds it, and the SyncTest's resimulated
    checksums then differ from the first ones -- for the oracle's callback exactly as for the device); Accel.z is arbitrary but never written."""
    D, ticks = 4, 14
    out = {}
    for strategy in (True, False):
        res = []
        for w in (bg.World(n, max_depth=D + 1), OracleWorld(n, D + 1, FLAT)):
            A, C_ = _f16_world(w, strategy)
            rng = np.random.default_rng(5)
            x = (rng.integers(-40, 40, n) * 0.5).astype(np.float32); y = rng.uniform(1, 2, n).astype(np.float32); z = rng.uniform(-1000, 1000, n).astype(np.float32)
            w.spawn(n, {A: [cm.f32bits(x), cm.f32bits(y), cm.f32bits(z)], C_: [np.zeros(n, dtype=np.uint32)]})
            w.set_depth(D + 1)
            cs = []
            if isinstance(w, bg.World): w.profile_enable(True)
            F = 0
            for t in range(ticks):                                          # SyncTest-shaped lists, without the session's mismatch check (a lossy strategy MUST mismatch)
                reqs = []
                if F >= D:
                    reqs.append(bg.LoadGameState(F - D))
                    for i in range(D): reqs += [bg.AdvanceFrame((0,))] + ([bg.SaveGameState(F - D + 1 + i)] if i < D - 1 else [])
                reqs += [bg.SaveGameState(F), bg.AdvanceFrame((0,))]
                w.set_confirmed(max(0, F - D))
                cs += w.handle_requests(reqs)
                F += 1
            if isinstance(w, bg.World):
                out[("bytes", strategy)] = w.profile_bytes()["tick"] / max(1, w.profile_read()["tick"][1]); w.profile_enable(False)
            res.append((cs, cm.snapshot_state(w, (A, C_))))
        assert res[0][0] == res[1][0], f"strategy={strategy}"
        cm.assert_states_equal(res[0][1], res[1][1], f"strategy={strategy}")
        out[strategy] = res[0]
    # x is a bijection under f16, y is not: the lossy world's checksums differ from the plain world's once a LoadWorld has happened
    assert out[True][0][:D] == out[False][0][:D] and out[True][0] != out[False][0]
    xs = [np.array(out[s][1]["c0w0"]) for s in (True, False)]
    assert np.array_equal(xs[0], xs[1])                                       # the exactly representable word is unaffected
    assert out[("bytes", True)] < out[("bytes", False)]                       # Saves move 6 B of Accel instead of 12 (ggrs_hip_profile_read_bytes counts the Stored form)


# ------------------------------------------------------------------------------------------------ timeline + shipped code objects
def test_host_timeline_counts_where_the_host_spends_a_tick():
    n, D = 300_000, 8
    w = bg.World(n, max_depth=D + 1)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(D + 1); w.set_synctest_check_distance(D)
    for _ in range(D + 1): w.handle_requests([bg.SaveGameState(w.frame), bg.AdvanceFrame((0,))])
    def tick():
        F = w.frame
        reqs = [bg.LoadGameState(F - D), bg.AdvanceFrame((0,))]
        for k in range(1, D + 1): reqs += [bg.SaveGameState(F - D + k), bg.AdvanceFrame((0,))]
        return reqs
    w.host_timeline(1)
    w.enqueue_requests(tick())
    for _ in range(19):
        w.enqueue_requests(tick()); w.collect_checksums()
    w.collect_checksums()
    t = w.host_timeline(0)
    assert t["enqueue_calls"] == 20 and t["collect_calls"] == 20 and 20 <= t["launches"] <= 21, t       # (+ one k_ff_fold-less flush at most)
    assert 0 < t["launch_call_us"] < t["enqueue_us"] and t["validate_us"] < t["enqueue_us"], t
    assert t["event_wait_us"] + t["tag_wait_us"] + t["host_fold_us"] <= t["collect_us"] * 1.01, t
    assert t["tag_wait_us"] > 0 and t["host_fold_us"] / 20 < 10.0, t                                     # fold-forward: the host hashes a few values per tick, it does not fold rows
    w.close()


def _hiprtc_compile(src: str) -> bytes:
    rtc = C.CDLL("libhiprtc.so")
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-ffp-contract=off", b"-fno-fast-math", b"-fhip-fp32-correctly-rounded-divide-sqrt"]
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"k.hip", 0, None, None) == 0
    assert rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts)) == 0
    n = C.c_size_t(); rtc.hiprtcGetCodeSize(prog, C.byref(n)); code = C.create_string_buffer(n.value); rtc.hiprtcGetCode(prog, code)
    return code.raw


def test_shipped_code_objects_serve_a_world_without_the_runtime_compiler(tmp_path, monkeypatch):
    """`make aot` in miniature: the generic and the steady text of a world shape compiled ahead of time, put under GGRS_AOT_DIR by the
    names the library asks for, and the same world run with the run-time compiler treated as absent (GGRS_NO_HIPRTC=1): both kernels come
    from the shipped objects, the session is the oracle's bit for bit."""
    from bevy_ggrs_amd import _ffi
    n, D = 123_456, 6                                                        # a shape no other test uses (the in-process module cache must not serve it)
    dry = bg.World(n, max_depth=D + 1, flags=bg.GGRS_WORLD_LAYOUT_ONLY)
    cm.build_particles(dry)
    for steady in (False, True):
        src = dry.generated_kernel_source(steady=steady)
        name = C.create_string_buffer(64)
        assert _ffi.lib.ggrs_hip_aot_object_name(src.encode(), name, 64) == 0
        (tmp_path / name.value.decode()).write_bytes(_hiprtc_compile(src))
    monkeypatch.setenv("GGRS_AOT_DIR", str(tmp_path)); monkeypatch.setenv("GGRS_NO_HIPRTC", "1"); monkeypatch.setenv("GGRS_JIT_CACHE_DIR", "0")
    res = []
    for w in (bg.World(n, max_depth=D + 1), OracleWorld(n, D + 1, FLAT)):
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        drv = cm.SyncTestDriver(w, D, max_prediction=D + 1)
        for _ in range(40): drv.tick((0,))
        if isinstance(w, bg.World):
            w.specialise_wait()
            for _ in range(6): drv.tick((0,))
            info = w.kernel_info()
            assert info["hiprtc"].startswith("missing") and info["generated_kernel"] == "ok" and "shipped" in info["generated_kernel_origin"], info
            assert info["request_group_kernel"].startswith("ggrs_jit_tick"), info
            assert info["specialised_kernel"].startswith("ready"), info      # the steady tick's copy was found among the shipped objects too
        else:
            for _ in range(6): drv.tick((0,))
        res.append((drv.all_checksums, cm.snapshot_state(w, ids)))
    assert res[0][0] == res[1][0]
    cm.assert_states_equal(res[0][1], res[1][1], "aot")


# ------------------------------------------------------------------------------------------------ lazy live block
def _lazy_counts(w):
    import re
    m = re.search(r"(\d+) lists left it unwritten, (\d+) materialised", w.kernel_info().get("lazy_live_block", ""))
    return (int(m.group(1)), int(m.group(2))) if m else None


@pytest.mark.parametrize("enqueue", [False, True])
def test_lazy_live_block_is_materialised_for_whoever_reads_it(enqueue):
    """An HBM-sized session whose lists keep opening with a LoadGameState stops writing the live block at the end of a tick (nobody reads it:
    the next LoadWorld replaces it).  Everything that DOES read or edit its bytes gets them first: a download, a host-side spawn, a list that
    opens without a Load (a P2P tick with no rollback), a despawn.  Checksums, snapshots and the final world are the oracle's throughout."""
    n, D = 450_000, 3
    cap = n + 64
    lib, orc = bg.World(cap, max_depth=D + 1), OracleWorld(cap, D + 1, FLAT)
    st = []
    for w in (lib, orc):
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        w.set_depth(D + 1)
        st.append({"w": w, "ids": ids, "F": 0, "cs": []})

    def run(s, reqs):
        w = s["w"]
        if enqueue and isinstance(w, bg.World):
            w.enqueue_requests(reqs); s["cs"] += w.collect_checksums()
        else:
            s["cs"] += w.handle_requests(reqs)

    def synctest_tick(s):
        F = s["F"]; reqs = []
        s["w"].set_confirmed(max(0, F - D))
        if F >= D:
            reqs.append(bg.LoadGameState(F - D))
            for i in range(D): reqs += [bg.AdvanceFrame((0,))] + ([bg.SaveGameState(F - D + 1 + i)] if i < D - 1 else [])
        reqs += [bg.SaveGameState(F), bg.AdvanceFrame((0,))]
        run(s, reqs); s["F"] += 1

    def plain_tick(s):                                                  # no rollback this tick: the list reads the live world
        run(s, [bg.SaveGameState(s["F"]), bg.AdvanceFrame((0,))]); s["F"] += 1

    def both(fn, k=1):
        for _ in range(k):
            for s in st: fn(s)

    def same(ctx):
        assert st[0]["cs"] == st[1]["cs"], ctx
        cm.assert_states_equal(cm.snapshot_state(lib, st[0]["ids"]), cm.snapshot_state(orc, st[1]["ids"]), ctx)

    both(synctest_tick, D + 12)
    skips, mats = _lazy_counts(lib)
    assert skips >= 3 and mats == 0, (skips, mats)                      # the streak of Load-opening lists is long enough: the live block is no longer written
    assert lib.active_count() == orc.active_count()                     # ... until somebody asks
    assert _lazy_counts(lib)[1] == 1
    same("after the first materialisation")
    both(synctest_tick, 3)
    assert _lazy_counts(lib)[0] > skips                                 # (the streak is not broken by reading the world)
    # a host-side spawn edits the live world between two lists
    for s in st:
        w, ids = s["w"], s["ids"]
        vel, ttl = cm.synthetic_particles(16, ttl="throughput", seed=7)
        cm.spawn_particles(w, ids, 16, vel, ttl)
    both(plain_tick, 2)                                                 # snapshots of the grown world (the ring's older frames predate the spawn)
    same("after a host-side spawn")
    both(synctest_tick, LAZY := 10)
    m0 = _lazy_counts(lib)[1]
    both(plain_tick)                                                    # opens with a Save of the live world the last tick did not write
    assert _lazy_counts(lib)[1] == m0 + 1
    both(synctest_tick, 9)
    for s in st: s["w"].despawn(5)
    both(synctest_tick, 2)
    same("at the end")
    lib.close()


def test_lazy_live_block_is_off_where_the_live_world_holds_more_than_the_snapshots():
    """A handed-out column pointer or a world that fits the caches: every tick writes the live block."""
    for n, hand_out in ((450_000, True), (100_000, False)):
        D = 3
        w = bg.World(n, max_depth=D + 1)
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        w.set_depth(D + 1)
        if hand_out: w.column_device_ptr(ids[0], 0)
        F = 0
        for _ in range(D + 12):
            reqs = []
            if F >= D:
                reqs.append(bg.LoadGameState(F - D))
                for i in range(D): reqs += [bg.AdvanceFrame((0,))] + ([bg.SaveGameState(F - D + 1 + i)] if i < D - 1 else [])
            reqs += [bg.SaveGameState(F), bg.AdvanceFrame((0,))]
            w.handle_requests(reqs); F += 1
        c = _lazy_counts(w)
        assert c is None or c[0] == 0, (n, hand_out, w.kernel_info().get("lazy_live_block"))
        w.close()


# ------------------------------------------------------------------------------------------------ the component limit
@pytest.mark.parametrize("per_request,n,nc", [(False, 5000, 16), (True, 5000, 16), (False, 300_000, 16), (False, 5000, 32), (True, 5000, 32), (False, 300_000, 32)])
def test_worlds_at_the_component_limit_every_presence_mask_travels(per_request, n, nc):
    """16 components + the liveness mask are 17 masks of 16 words per 1024-slot tile = 272 words: the per-request copy kernel moved the first 256 (one trip of
    its 256 threads), i.e. a SaveWorld / LoadWorld lost the LAST component's presence mask (the limit was 16 then; GGRS_MAX_COMPONENTS is 32 now: 33 masks).
    Both paths against the oracle, with presence edits between the lists so that a LoadWorld has to bring the old masks back.  300 k: every component
    checksummed through the wave fold and, enqueued, the fold-forward role (nc + 1 values per Save and workgroup)."""
    res = []
    last = nc - 1
    for w in (bg.World(n, max_depth=4, flags=bg.GGRS_WORLD_NO_GROUPS if per_request else 0), OracleWorld(n, 4, FLAT)):
        comps = [w.register_component(f"C{k}", 4, 1) for k in range(nc)]
        for c in comps: w.checksum_component(c, [0])
        w.add_system(bg.SYS_ADD_U32, comp=(comps[last],), word=(0,), iparam=(3,))
        w.add_system(bg.SYS_ADD_U32, comp=(comps[0],), word=(0,), iparam=(1,))
        per = n // nc
        for k in range(nc):                                                     # batch k carries components k and (k + 5) % nc
            a, b = comps[k], comps[(k + 5) % nc]
            w.spawn(per, {a: [np.arange(per, dtype=np.uint32) + 100 * k], b: [np.full(per, 7 + k, dtype=np.uint32)]})
        w.set_depth(4)
        cs = []
        cs += w.handle_requests([bg.SaveGameState(0), bg.AdvanceFrame((0,)), bg.SaveGameState(1), bg.AdvanceFrame((0,))])
        w.remove_component(comps[last], last * per + 3)                         # live edits after frame 1's snapshot
        w.insert_component(comps[last], 2, np.array([41], dtype=np.uint32))
        cs += w.handle_requests([bg.SaveGameState(2), bg.AdvanceFrame((0,)), bg.SaveGameState(3), bg.AdvanceFrame((0,))])
        mid = cm.snapshot_state(w, comps)
        roll = [bg.LoadGameState(1), bg.AdvanceFrame((0,)), bg.SaveGameState(2), bg.AdvanceFrame((0,)), bg.SaveGameState(3), bg.AdvanceFrame((0,))]
        if isinstance(w, bg.World) and not per_request:
            w.enqueue_requests(roll); w.enqueue_requests([bg.LoadGameState(3), bg.AdvanceFrame((0,)), bg.SaveGameState(4), bg.AdvanceFrame((0,))])     # (the second list folds the first one's rows at 300 k)
            cs += w.collect_checksums(); cs += w.collect_checksums()
        else:
            cs += w.handle_requests(roll); cs += w.handle_requests([bg.LoadGameState(3), bg.AdvanceFrame((0,)), bg.SaveGameState(4), bg.AdvanceFrame((0,))])
        res.append((cs, mid, cm.snapshot_state(w, comps)))
    assert res[0][0] == res[1][0]
    cm.assert_states_equal(res[0][1], res[1][1], "before the rollback")
    cm.assert_states_equal(res[0][2], res[1][2], "after the rollback")
    assert res[0][1][f"present{last}"][2] and not res[0][2][f"present{last}"][2]     # the rollback took the inserted component away again
    assert res[0][2][f"present{last}"][last * per + 3]                               # ... and brought the removed one back
