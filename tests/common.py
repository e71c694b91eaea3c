"""Shared fixtures: the stress_test (particles) world of examples/stress_tests/particles.rs
built on either backend, the synthetic inputs of BASELINE.md section 3, and a backend-neutral
SyncTest driver (run_synctest, src/schedule_systems.rs:85-118)."""
from __future__ import annotations

import numpy as np

import bevy_ggrs_amd as bg
from bevy_ggrs_amd.session import SyncTestSession

INPUT_SPAWN = 1 << 4           # particles.rs:75
TRANSFORM_DEFAULT = np.array([0, 0, 0, 0, 0, 0, 1, 1, 1, 1], dtype=np.float32)  # t(3) rot xyzw(4) scale(3)


def f32bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# examples/stress_tests/particles.rs:190-199 registers Sprite (not POD: it holds an asset handle -- out of scope), Transform,
# GlobalTransform, Visibility, InheritedVisibility, ViewVisibility, Velocity, Ttl.  "full" = that list minus Sprite:
# GlobalTransform is an Affine3A (3x3 matrix + translation = 12 f32), Visibility a 1-byte enum, the other two 1-byte bools.
GLOBAL_TRANSFORM_DEFAULT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)   # GlobalTransform::IDENTITY
FULL_EXTRA = (("GlobalTransform", 4, 12), ("Visibility", 1, 1), ("InheritedVisibility", 1, 1), ("ViewVisibility", 1, 1))


def schema_bytes_per_entity(schema="headline"):
    return 60 if schema in ("headline", "allhot") else 60 + sum(wb * nw for _, wb, nw in FULL_EXTRA)


def schema_description(schema="headline"):
    if schema == "headline":
        return "3 registered components (Transform, Velocity, Ttl; 60 B/entity)"
    if schema == "allhot":
        return ("3 registered components (Transform, Velocity, Ttl; 60 B/entity) with EVERY column written every frame: the stress_test systems plus the "
                "reference bench's increase_component (benches/bench.rs:30-46) over the 7 words of Transform.rotation / .scale -- every SaveWorld moves all 15 rows, "
                "as the reference's clone-everything save does (component_snapshot.rs:66-84)")
    return (f"7 registered components (Transform, GlobalTransform, Visibility, InheritedVisibility, ViewVisibility, Velocity, Ttl; "
            f"{schema_bytes_per_entity(schema)} B/entity: the reference stress_test's rollback list minus Sprite, particles.rs:190-199)")


def build_particles(world, *, with_spawn=False, ttl_init=300, checksum=True, schema="headline"):
    """particles.rs:187-240: the 3 registered components of SURVEY.md section 8 ("headline"), or the reference's whole POD
    rollback list ("full": the extra components are registered FIRST-TO-LAST as the reference does, ids 3..6, and returned
    after T, V, L so that callers indexing ids[:3] keep working)."""
    T = world.register_component("Transform", 4, 10)
    V = world.register_component("Velocity", 4, 3)
    L = world.register_component("Ttl", 8, 1)
    extra = ()
    if schema == "full":
        extra = tuple(world.register_component(nm, wb, nw) for nm, wb, nw in FULL_EXTRA)
        world.set_component_default(extra[0], GLOBAL_TRANSFORM_DEFAULT)
        for c, v in zip(extra[1:], (0, 1, 0)):                  # Visibility::Inherited, InheritedVisibility(true), ViewVisibility(false)
            world.set_component_default(c, np.array([v], dtype=np.uint8))
    world.set_component_default(T, TRANSFORM_DEFAULT)
    if checksum:
        world.checksum_component(V, [0, 1, 2])        # checksum_component_with_hash::<Velocity>()
        world.checksum_component(T, [0, 1, 2])        # checksum_component::<Transform>(translation)
    world.add_system(bg.SYS_PARTICLES_UPDATE, comp=(T, V), word=(0, 0), fparam=(0.0, -200.0, 0.0))
    world.add_system(bg.SYS_TTL_DESPAWN, comp=(L,), word=(0,))
    if schema == "allhot":
        # a game whose systems touch rotation and scale too: no column keeps its bytes from one frame to the next, so row versions
        # cannot skip anything -- the all-columns-hot case of VERDICT r3 (What's missing 3)
        for k in range(3, 10):
            world.add_system(bg.SYS_ADD_U32, comp=(T,), word=(k,), iparam=(1,))
    if with_spawn:
        world.add_system(bg.SYS_PARTICLES_SPAWN, comp=(T, V, L), iparam=(ttl_init, INPUT_SPAWN))
    return (T, V, L) + extra


def synthetic_particles(n, ttl="throughput", seed=123):
    """Velocity=(u1,u2,0), u~U[-200,200) from default_rng(123); Transform default."""
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-200, 200, (n, 2)).astype(np.float32)
    if ttl == "throughput":
        t = np.full(n, 1 << 40, dtype=np.uint64)
    elif ttl == "despawn":
        t = (1 + (np.arange(n, dtype=np.uint64) % 300)).astype(np.uint64)
    else:
        t = np.full(n, int(ttl), dtype=np.uint64)
    return vel, t


def spawn_particles(world, ids, n, vel, ttl):
    T, V, L = ids[:3]
    tcols = [np.full(n, f32bits(TRANSFORM_DEFAULT)[k], dtype=np.uint32) for k in range(10)]
    vcols = [f32bits(vel[:, 0]), f32bits(vel[:, 1]), np.zeros(n, dtype=np.uint32)]
    bundle = {T: tcols, V: vcols, L: [ttl]}
    for c in ids[3:]: bundle[c] = None                           # the extra components of the "full" schema: their defaults
    return world.spawn(n, bundle)


def snapshot_state(world, ids):
    """Everything observable: (len, alive, per-component (present&alive, words of live entities))."""
    n = world.len
    alive = world.alive_mask(n)
    out = {"len": n, "frame": world.frame, "alive": alive}
    for cid in ids:
        _, wb, nw = world._comps[cid]
        pres = world.present_mask(cid, n) & alive
        out[f"present{cid}"] = pres
        for k in range(nw):
            col = world.download_word(cid, k, 0, n)
            out[f"c{cid}w{k}"] = np.where(pres, col, 0)
    return out


def assert_states_equal(a, b, ctx=""):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            if not np.array_equal(a[k], b[k]):
                bad = np.nonzero(a[k] != b[k])[0]
                raise AssertionError(f"{ctx} state field {k} differs at {bad[:8]} ({bad.size} slots): {a[k][bad[:4]]} vs {b[k][bad[:4]]}")
        else:
            assert a[k] == b[k], (ctx, k, a[k], b[k])


class SyncTestDriver:
    """run_synctest + handle_requests over any backend world (schedule_systems.rs:85-118,170-289)."""

    def __init__(self, world, check_distance, num_players=1, max_prediction=None, input_delay=0):
        self.world = world
        mp = max_prediction if max_prediction is not None else max(8, check_distance + 1)
        self.sess = SyncTestSession(num_players, check_distance, mp, input_delay)
        self.cd = check_distance
        self.lib_rule = hasattr(world, "_lib") and world._prefix == "ggrs_hip_"
        world.set_depth(mp)                       # sync_depth: MaxPredictionWindow (mod.rs:263-273)
        if self.lib_rule:
            world.set_synctest_check_distance(check_distance)
        self.all_checksums = []                   # (frame, checksum) of every save, in order

    def _confirm_rule(self):
        # schedule_systems.rs:204-220 (applied per request; the oracle has no built-in rule)
        c = self.world.frame - self.cd
        if c >= 0:
            self.world.set_confirmed(c)

    def tick(self, inputs=(0,), spawn_fn=None, patch=None):
        """spawn_fn(frame) -> (vx, vy): the ParticleRng draw of the frame being advanced.  It must
        be a pure function of the frame: ParticleRng is a rollback resource
        (particles.rs:201), so a resimulated frame redraws the same values.
        patch(frame, advance_request): any other per-frame decoration of an AdvanceFrame (spawn_count / spawn_payload of a
        user-written spawn system, InputStatus, wider inputs) -- a pure function of the frame and its inputs for the same reason."""
        for h, v in enumerate(inputs):
            self.sess.add_local_input(h, v)
        reqs = self.sess.advance_frame()
        if spawn_fn is not None or patch is not None:
            cur = self.world.frame
            for r in reqs:
                if isinstance(r, bg.LoadGameState):
                    cur = r.frame
                elif isinstance(r, bg.AdvanceFrame):
                    if spawn_fn is not None and any(i & INPUT_SPAWN for i in r.inputs):
                        r.spawn_vx, r.spawn_vy = spawn_fn(cur)
                    if patch is not None: patch(cur, r)
                    cur += 1
        if self.lib_rule:
            cs = self.world.handle_requests(reqs)
        else:
            cs = []
            for r in reqs:
                self._confirm_rule()
                cs += self.world.handle_requests([r])
        self.sess.record_checksums(cs)
        saves = [r for r in reqs if isinstance(r, bg.SaveGameState)]
        self.all_checksums += [(s.frame, c) for s, c in zip(saves, cs)]
        return cs


def frame_spawn_fn(rate=100, seed=99):
    """Deterministic per-frame spawn payload (stands in for the rolled-back ParticleRng)."""
    def fn(frame):
        r = np.random.default_rng([seed, frame])
        return (r.uniform(-200, 200, rate).astype(np.float32), r.uniform(-200, 200, rate).astype(np.float32))
    return fn


class P2PShapeDriver:
    """Synthetic stand-in for ggrs's P2PSession::advance_frame on BASELINE config 4 (no sockets): every tick a
    late remote input invalidates the last `r` predicted frames (r drawn per tick, 0..max_rollback; 120 ms RTT at
    60 fps ~ 7-8 frames), so the request list is  [Load(F-r), Adv, (Save, Adv) x (r-1)]  +  [Save(F), Adv]
    (non-sparse saving; shape restated from ggrs) and ConfirmedFrameCount trails by max_rollback
    (schedule_systems.rs:202 `s.confirmed_frame()`).  inputs(frame) must be a pure function of the frame."""

    def __init__(self, world, max_rollback=8, seed=4, inputs=lambda frame: (0,), spawn_fn=None):
        self.world, self.max_rollback = world, max_rollback
        self.rng = np.random.default_rng(seed)
        self.inputs, self.spawn_fn = inputs, spawn_fn
        self.frame = world.frame
        self.lib_rule = hasattr(world, "_lib") and world._prefix == "ggrs_hip_"
        world.set_depth(max_rollback)
        if self.lib_rule:
            world.set_synctest_check_distance(-1)
        self.all_checksums = []
        self.depths = []

    def _adv(self, frame):
        a = bg.AdvanceFrame(tuple(self.inputs(frame)))
        if self.spawn_fn is not None and any(i & INPUT_SPAWN for i in a.inputs):
            a.spawn_vx, a.spawn_vy = self.spawn_fn(frame)
        return a

    def tick(self):
        F = self.frame
        r = int(self.rng.integers(0, self.max_rollback + 1))
        r = min(r, F, self.max_rollback - 1 if self.max_rollback > 1 else 0)   # only frames still in the ring
        reqs = []
        if r > 0:
            reqs.append(bg.LoadGameState(F - r))
            for i in range(r):
                if i > 0:
                    reqs.append(bg.SaveGameState(F - r + i))
                reqs.append(self._adv(F - r + i))
        reqs += [bg.SaveGameState(F), self._adv(F)]
        c = F - self.max_rollback
        if c >= 0:
            self.world.set_confirmed(c)
        cs = self.world.handle_requests(reqs)
        saves = [q for q in reqs if isinstance(q, bg.SaveGameState)]
        self.all_checksums += [(s.frame, x) for s, x in zip(saves, cs)]
        self.depths.append(r)
        self.frame = F + 1
        return cs
