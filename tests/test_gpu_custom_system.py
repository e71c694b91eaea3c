"""User-written GgrsSchedule systems (ggrs_hip_add_custom_system, GGRS_SYS_CUSTOM): the reference accepts ANY Bevy
system in GgrsSchedule (src/lib.rs:76, 247-251); the HIP engine accepts a per-entity system as HIP C++ source and
compiles it for gfx950 with hiprtc.  Parity is pinned by writing the reference's own systems a second time as custom
source and requiring the world that runs them to match, bit for bit, the CPU oracle running its built-in restatement:
    update_particles / despawn_particles   examples/stress_tests/particles.rs:272-289
    increase_component (benches)           benches/bench.rs:30-46 shape
    decrease_health + despawn_rollback     tests/synctest.rs:37-44, src/snapshot/despawn.rs:114-143
"""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu

UPDATE_SRC = r"""
// particles.rs:272-280: velocity += gravity * dt; translation += velocity * dt   (bindings: T.xyz, V.xyz)
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {
    for (int k = 0; k < 3; ++k) {
        e.f32(3 + k) += f.fparam[k] * f.dt;
        e.f32(k) += e.f32(3 + k) * f.dt;
    }
}
"""
TTL_SRC = r"""
// particles.rs:282-289: ttl -= 1 (usize, wrapping); == 0 -> despawn
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) {
    e.u64(0) -= 1;
    if (e.u64(0) == 0) e.despawn();
}
"""
ADD_SRC = r"""
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) { e.u32(0) += (unsigned)f.iparam[0]; }
"""
HEALTH_SRC = r"""
// tests/synctest.rs:37-44 with commands.entity(e).despawn_rollback() / despawn() chosen by iparam[1]
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {
    const unsigned a = (unsigned)f.iparam[0];
    e.u32(0) = e.u32(0) >= a ? e.u32(0) - a : 0u;
    if (e.u32(0) == 0) { if (f.iparam[1]) e.despawn_rollback(); else e.despawn(); }
}
"""


def build_particles_custom(world, *, with_spawn, ttl_init):
    T = world.register_component("Transform", 4, 10)
    V = world.register_component("Velocity", 4, 3)
    L = world.register_component("Ttl", 8, 1)
    world.set_component_default(T, cm.TRANSFORM_DEFAULT)
    world.checksum_component(V, [0, 1, 2])
    world.checksum_component(T, [0, 1, 2])
    world.add_custom_system(UPDATE_SRC, [(T, 0), (T, 1), (T, 2), (V, 0), (V, 1), (V, 2)], fparam=(0.0, -200.0, 0.0), name="update_particles")
    world.add_custom_system(TTL_SRC, [(L, 0)], name="despawn_particles")
    if with_spawn:
        world.add_system(bg.SYS_PARTICLES_SPAWN, comp=(T, V, L), iparam=(ttl_init, cm.INPUT_SPAWN))   # built-in and custom kinds mix
    return T, V, L


def _drive(world, ids, n, cd, ticks, with_spawn):
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, cd)
    fn = cm.frame_spawn_fn(100)
    for t in range(ticks):
        drv.tick((cm.INPUT_SPAWN if (with_spawn and t % 3 == 1) else 0,), spawn_fn=fn)
    return drv.all_checksums, cm.snapshot_state(world, ids)


@pytest.mark.parametrize("n,cd,ticks,with_spawn", [(1, 2, 10, False), (1000, 2, 20, True), (8192 + 77, 7, 24, True), (100_000, 8, 12, False)])
def test_particles_written_as_custom_systems_match_the_oracle(n, cd, ticks, with_spawn):
    cap = n + 100 * ticks + 64
    g = bg.World(cap, max_depth=16)
    o = OracleWorld(cap, 16, FLAT)
    a = _drive(g, build_particles_custom(g, with_spawn=with_spawn, ttl_init=40), n, cd, ticks, with_spawn)
    b = _drive(o, cm.build_particles(o, with_spawn=with_spawn, ttl_init=40), n, cd, ticks, with_spawn)
    assert len(a[0]) == len(b[0]) > 0
    for (fa, ca), (fb, cb) in zip(a[0], b[0]):
        assert fa == fb and ca == cb, f"frame {fa}: gpu {ca:#x} oracle {cb:#x}"
    cm.assert_states_equal(a[1], b[1], f"custom particles n={n}")


def test_custom_and_builtin_worlds_are_identical_on_the_gpu():
    """Same world twice on the GPU: built-in kinds (one fused launch per tick) vs custom source (request by request)."""
    n, cd, ticks = 50_000, 8, 16
    g1 = bg.World(n + 64, max_depth=16)
    g2 = bg.World(n + 64, max_depth=16)
    a = _drive(g1, cm.build_particles(g1, with_spawn=False), n, cd, ticks, False)
    b = _drive(g2, build_particles_custom(g2, with_spawn=False, ttl_init=40), n, cd, ticks, False)
    assert a[0] == b[0]
    cm.assert_states_equal(a[1], b[1], "builtin vs custom")


@pytest.mark.parametrize("n", [5, 4097])
def test_add_u32_custom_with_params_inputs_and_u64_words(n):
    g = bg.World(n, max_depth=8)
    o = OracleWorld(n, 8, FLAT)
    for w in (g, o):
        A = w.register_component("A", 4, 2)
        w.checksum_component(A, [0, 1])
    g.add_custom_system(ADD_SRC, [(0, 1)], iparam=(7,), name="increase")
    o.add_system(bg.SYS_ADD_U32, comp=(0,), word=(1,), iparam=(7,))
    cols = [np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32) * 3]
    out = []
    for w in (g, o):
        w.spawn(n, {0: cols})
        drv = cm.SyncTestDriver(w, 3)
        for _ in range(9): drv.tick((0,))
        out.append((drv.all_checksums, cm.snapshot_state(w, (0,))))
    assert out[0][0] == out[1][0]
    cm.assert_states_equal(out[0][1], out[1][1], "add_u32 custom")


def test_frame_view_dt_frame_inputs_and_slot():
    """GgrsFrame / GgrsEntity fields: f.dt bits, f.frame, f.input, e.slot land where the source puts them."""
    n = 200
    g = bg.World(n, max_depth=4)
    R = g.register_component("Rec", 4, 4)
    S = g.register_component("Slot", 8, 1)
    g.add_custom_system(r"""
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {
    e.f32(0) = f.dt; e.i32(1) = f.frame; e.u32(2) = f.n_inputs * 1000u + f.input[0] + 2u * f.input[1]; e.f32(3) = f.fparam[3];
    e.u64(4) = e.slot * 3ull + (unsigned long long)f.iparam[1];
}
""", [(R, 0), (R, 1), (R, 2), (R, 3), (S, 0)], iparam=(0, 11), fparam=(0, 0, 0, 2.5))
    g.spawn(n, {R: None, S: None})
    g.handle_requests([bg.AdvanceFrame((5, 9))])
    g.handle_requests([bg.AdvanceFrame((6, 1))])
    assert g.frame == 2
    dt = g.download_word(R, 0, 0, n).view(np.float32)
    assert np.all(dt == np.float32(1.0 / 60.0))                     # RollbackFrameRate default 60 (lib.rs:62)
    assert np.all(g.download_word(R, 1, 0, n).view(np.int32) == 2)
    assert np.all(g.download_word(R, 2, 0, n) == 2000 + 6 + 2)
    assert np.all(g.download_word(R, 3, 0, n).view(np.float32) == np.float32(2.5))
    assert np.array_equal(g.download_word(S, 0, 0, n), np.arange(n, dtype=np.uint64) * 3 + 11)


def test_system_only_sees_entities_with_every_bound_component():
    n = 300
    g = bg.World(2 * n, max_depth=4)
    A = g.register_component("A", 4, 1)
    B = g.register_component("B", 4, 1)
    g.add_custom_system("__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) { e.u32(0) += e.u32(1); }", [(A, 0), (B, 0)])
    g.spawn(n, {A: [np.full(n, 1, np.uint32)], B: [np.full(n, 10, np.uint32)]})
    g.spawn(n, {A: [np.full(n, 1, np.uint32)]})                      # no B: Query<(&mut A, &B)> skips them
    g.handle_requests([bg.AdvanceFrame((0,))])
    a = g.download_word(A, 0, 0, 2 * n)
    assert np.all(a[:n] == 11) and np.all(a[n:] == 1)


@pytest.mark.parametrize("mode", [bg.DESPAWN_IMMEDIATE, bg.DESPAWN_ROLLBACK])
def test_despawn_rollback_from_custom_source_matches_the_oracle(mode):
    """The scripted RollbackDespawned scenario of tests/test_despawn_rollback.py with decrease_health as custom source."""
    import test_despawn_rollback as dr
    n = 300

    def build_custom(world, n_, mode_=mode, checksum=True):
        H = world.register_component("Health", 4, 1)
        M = world.register_component("Mesh", 4, 2, rollback=False)
        world.checksum_component(H, [0])
        world.add_custom_system(HEALTH_SRC, [(H, 0)], iparam=(1, mode_), name="decrease_health")
        health = (1 + (np.arange(n_) % 5)).astype(np.uint32)
        world.spawn(n_, {H: [health], M: [np.arange(n_, dtype=np.uint32) + 1000, np.arange(n_, dtype=np.uint32) * 7]})
        return H, M

    g = bg.World(n, max_depth=8)
    o = OracleWorld(n, 8, FLAT)
    saved = dr.build
    try:
        dr.build = build_custom
        a = dr.scripted(g, n)
        dr.build = lambda w, n_: saved(w, n_, mode)
        b = dr.scripted(o, n)
    finally:
        dr.build = saved
    assert [k for k, _ in a] == [k for k, _ in b]
    for (k, sa), (_, sb) in zip(a, b):
        for key in sb:
            assert np.array_equal(np.asarray(sa[key]), np.asarray(sb[key])), (mode, k, key)


def test_live_only_component_can_be_bound():
    """A system may touch a component that is NOT registered for rollback (plain Bevy component): it is stepped but never
    saved or restored -- LoadWorld leaves it as the last simulated frame wrote it."""
    n = 9000
    g = bg.World(n, max_depth=4)
    H = g.register_component("Health", 4, 1)
    K = g.register_component("Counter", 4, 1, rollback=False)
    g.add_custom_system("__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) { e.u32(0) += 1; e.u32(1) += 1; }", [(H, 0), (K, 0)])
    g.spawn(n, {H: [np.zeros(n, np.uint32)], K: [np.arange(n, dtype=np.uint32)]})
    g.handle_requests([bg.SaveGameState(0), bg.AdvanceFrame((0,)), bg.AdvanceFrame((0,)), bg.LoadGameState(0), bg.AdvanceFrame((0,))])
    assert np.all(g.download_word(H, 0, 0, n) == 1)                                  # rolled back, then one frame
    assert np.array_equal(g.download_word(K, 0, 0, n), np.arange(n, dtype=np.uint32) + 3)   # three simulated frames, no rollback


def test_compile_error_is_reported_with_the_compiler_log_and_the_world_stays_usable():
    g = bg.World(64, max_depth=4)
    A = g.register_component("A", 4, 1)
    with pytest.raises(bg.GgrsHipError) as ei:
        g.add_custom_system("__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) { e.u32(0) += undefined_symbol; }", [(A, 0)], name="broken")
    assert ei.value.code == bg.GGRS_E_INVALID
    assert "broken" in str(ei.value) and "undefined_symbol" in str(ei.value)
    with pytest.raises(bg.GgrsHipError):                                             # wrong signature: no ggrs_system to call
        g.add_custom_system("__device__ void other() {}", [(A, 0)])
    with pytest.raises(bg.GgrsHipError):
        g.add_custom_system(ADD_SRC, [(A, 5)])                                       # word out of range
    with pytest.raises(bg.GgrsHipError):
        g.add_custom_system(ADD_SRC, [(3, 0)])                                       # unknown component
    g.add_custom_system(ADD_SRC, [(A, 0)], iparam=(2,))
    g.spawn(64, {A: [np.zeros(64, np.uint32)]})
    g.handle_requests([bg.AdvanceFrame((0,))])
    assert np.all(g.download_word(A, 0, 0, 64) == 2)
    with pytest.raises(bg.GgrsHipError):
        g.add_custom_system(ADD_SRC, [(A, 0)])                                       # sealed


BOX_SRC = r"""
// examples/box_game/box_game.rs:154-206 move_cube_system, written by the "user": bindings T.xyz (0..2), V.xyz (3..5), Player.handle (6)
// fparam = {ACCELERATION, MAX_SPEED, FRICTION.powf(dt), half_width}.  sqrt and / must be the correctly rounded IEEE operations.
__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {
    const unsigned long long handle = e.u64(6);
    if (handle >= f.n_inputs) return;
    const unsigned char in = f.input[handle];
    const bool up = in & 1, down = in & 2, left = in & 4, right = in & 8;
    float x = e.f32(0), y = e.f32(1), z = e.f32(2), vx = e.f32(3), vy = e.f32(4), vz = e.f32(5);
    const float adt = f.fparam[0] * f.dt, fp = f.fparam[2], max_speed = f.fparam[1];
    if (up && !down) vz -= adt;
    if (!up && down) vz += adt;
    if (left && !right) vx -= adt;
    if (!left && right) vx += adt;
    if (!up && !down) vz *= fp;
    if (!left && !right) vx *= fp;
    vy *= fp;
    const float len_sq = vx * vx + vy * vy + vz * vz;
    if (len_sq > max_speed * max_speed) { const float l = sqrtf(len_sq); vx = max_speed * (vx / l); vy = max_speed * (vy / l); vz = max_speed * (vz / l); }
    x += vx * f.dt; y += vy * f.dt; z += vz * f.dt;
    const float lo = -f.fparam[3], hi = f.fparam[3];
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    if (z < lo) z = lo;
    if (z > hi) z = hi;
    e.f32(0) = x; e.f32(1) = y; e.f32(2) = z; e.f32(3) = vx; e.f32(4) = vy; e.f32(5) = vz;
}
"""


def test_box_game_written_as_a_custom_system_matches_the_oracle():
    """box_game's move_cube_system as user source (branches, sqrt, divide, clamps, PlayerInputs) == the oracle's built-in
    restatement, bit for bit, over a SyncTest with random inputs: the run-time compiler keeps the library's floating-point contract
    (no contraction, correctly rounded sqrt and divide).  FRICTION.powf(dt) is the platform libm's, as in the reference's build."""
    import ctypes
    import test_box_game as tb
    libm = ctypes.CDLL("libm.so.6"); libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    n, players, cd, ticks = 5000, 4, 7, 30
    # RollbackFrameRate(50): 20 000 000 ns per frame exactly, so every frame's delta (time.rs:63-87) is the same f32 and the
    # friction factor is one constant (at 60 fps the integer-nanosecond deltas alternate between two f32 values)
    dt = np.float32(20_000_000) / np.float32(1_000_000_000)
    fp = float(libm.powf(tb.BOX_PARAMS[2], float(dt)))
    out = []
    for custom in (True, False):
        w = bg.World(n, max_depth=9) if custom else OracleWorld(n, 9, FLAT)
        T = w.register_component("Transform", 4, 10); V = w.register_component("Velocity", 4, 3); P = w.register_component("Player", 8, 1)
        w.set_component_default(T, cm.TRANSFORM_DEFAULT)
        w.set_frame_rate(50)
        w.checksum_component(T, [0, 1, 2]); w.checksum_component(V, [0, 1, 2])
        if custom:
            w.add_custom_system(BOX_SRC, [(T, 0), (T, 1), (T, 2), (V, 0), (V, 1), (V, 2), (P, 0)],
                                fparam=(tb.BOX_PARAMS[0], tb.BOX_PARAMS[1], fp, tb.BOX_PARAMS[3]), name="move_cube_system")
        else:
            w.add_system(bg.SYS_BOX_MOVE, comp=(T, V, P), word=(0, 0, 0), fparam=tb.BOX_PARAMS)
        rng = np.random.default_rng(5)
        tr = np.tile(cm.TRANSFORM_DEFAULT, (n, 1)).astype(np.float32)
        tr[:, 0:3] = rng.uniform(-2.6, 2.6, (n, 3)).astype(np.float32)
        vel = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
        handle = (np.arange(n) % (players + 1)).astype(np.uint64)         # some handles have no input: inputs[handle] would panic
        w.spawn(n, {T: [cm.f32bits(tr[:, k]) for k in range(10)], V: [cm.f32bits(vel[:, k]) for k in range(3)], P: [handle]})
        drv = cm.SyncTestDriver(w, cd, num_players=players)
        for t in range(ticks):
            drv.tick(tb.input_script(t, players))
        out.append((drv.all_checksums, cm.snapshot_state(w, (T, V, P))))
    assert out[0][0] == out[1][0]
    cm.assert_states_equal(out[0][1], out[1][1], "box_game as custom source")
