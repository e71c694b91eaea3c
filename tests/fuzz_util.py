"""Differential fuzzing of the request path: seeded random worlds and random VALID request lists, run in lock-step on a
reference backend A (the CPU oracle: it also serves as the model the generator consults) and a backend under test B.

What a list may contain follows what `handle_requests` accepts (src/schedule_systems.rs:170-289): SaveGameState of the current
frame, LoadGameState of a frame that is still in the ring (GgrsSnapshots, src/snapshot/mod.rs:147-226: pushes evict beyond the
depth, a rollback pops everything newer, confirmation prunes everything older) and AdvanceFrame with any inputs -- in ANY order,
not only the shapes ggrs's sessions emit: SyncTest ticks, P2P-shaped rollbacks and speculative branch lists are generated on
purpose (they are what the fused request groups, dead-snapshot elimination and batches key on), everything else at random.
Between lists: ConfirmedFrameCount moves, entities are despawned, components removed and inserted from the host.

Compared after every list: the Checksum(u128) of every Save, frame, RollbackOrdered::len, the number of snapshots; every few
lists and at the end: every word of every component of every live entity, the liveness and presence masks, which frames the
ring still holds."""
from __future__ import annotations

from collections import deque

import numpy as np

import bevy_ggrs_amd as bg
import common as cm


def w32(x):
    """i32 arithmetic of the frame counters (Frame = i32; RollbackFrameCount wraps: schedule_systems.rs:223-268)."""
    return ((int(x) + 2**31) % 2**32) - 2**31


def ahead(a, b):
    """How many frames `a` is ahead of `b`, modulo 2^32."""
    return (int(a) - int(b)) % 2**32


class _Ring:
    """Which frames the snapshot ring holds (mod.rs:147-226), without confirmation: the generator only loads frames at or
    above ConfirmedFrameCount, which confirmation never prunes (mod.rs:185-202 pops frames BELOW it -- by a plain signed
    comparison, also across the i32 wrap, which this model therefore never relies on)."""

    def __init__(self, depth): self.frames, self.depth = deque(), depth

    def push(self, f):
        while self.frames:                                   # mod.rs:147-181: "newer" is decided wrap-aware (abs_diff > u32::MAX / 2)
            cur = self.frames[0]
            wrapped = abs(cur - f) > (2**32 - 1) // 2
            if (cur >= f and not wrapped) or (f >= cur and wrapped): self.frames.popleft()
            else: break
        self.frames.appendleft(f)
        while len(self.frames) > self.depth: self.frames.pop()

    def rollback(self, f):
        while self.frames[0] != f: self.frames.popleft()

    def confirm(self, c):
        """mod.rs:185-202, applied no later than the backends do (at every push and whenever ConfirmedFrameCount moves): across the i32
        wrap a frame pruned while ConfirmedFrameCount was a large positive number must not look loadable again once it has wrapped."""
        while self.frames and self.frames[-1] < c: self.frames.pop()


class Scenario:
    def __init__(self, seed, *, big=False, max_n=None):
        r = self.rng = np.random.default_rng([0xF022, seed])
        self.seed = seed
        sizes = [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 8191, 8192, 8193, 20_000] if not big else [300_000, 700_001]
        self.n = int(r.choice([s for s in sizes if not max_n or s <= max_n]))
        self.schema = str(r.choice(["headline", "headline", "allhot", "full"]))
        self.with_spawn = bool(r.random() < 0.6)
        self.rate = int(r.choice([1, 7, 64, 300])) if not big else int(r.choice([100, 5000]))
        if max_n: self.rate = min(self.rate, 7)
        self.spawn_budget = 24
        self.ttl_mode = str(r.choice(["despawn", "throughput", "short"]))
        self.ttl_init = int(r.choice([2, 5, 300]))
        self.depth = int(r.integers(1, 9))
        self.checksum = bool(r.random() < 0.9)
        self.capacity = self.n + self.spawn_budget * self.rate + 8
        self.spawn_fn = cm.frame_spawn_fn(self.rate, seed=1000 + seed)

    def describe(self):
        return (f"seed {self.seed}: n={self.n} schema={self.schema} spawn={self.with_spawn} (rate {self.rate}, ttl_init {self.ttl_init}) ttl={self.ttl_mode} "
                f"depth={self.depth} checksum={self.checksum}")

    def build(self, world):
        ids = cm.build_particles(world, with_spawn=self.with_spawn, ttl_init=self.ttl_init, checksum=self.checksum, schema=self.schema)
        vel, ttl = cm.synthetic_particles(self.n, ttl={"short": 9}.get(self.ttl_mode, self.ttl_mode), seed=self.seed)
        cm.spawn_particles(world, ids, self.n, vel, ttl)
        world.set_depth(self.depth)
        if hasattr(world, "set_synctest_check_distance"): world.set_synctest_check_distance(-1)      # ConfirmedFrameCount is set explicitly on both backends
        return ids


_DT = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


class GenericScenario:
    """A world of made-up components (1-, 2-, 4- and 8-byte words, some entities without some components, optionally one component
    that is not registered for rollback), add_u32 systems (benches/bench.rs:30-46) and optionally a Health-style countdown that
    despawns at zero, immediately or with a RollbackDespawned marker (tests/synctest.rs:37-44, snapshot/despawn.rs:114-143)."""
    with_spawn, spawn_budget, spawn_fn = False, 0, None

    def __init__(self, seed, *, big=False, max_n=None):
        r = self.rng = np.random.default_rng([0x6E6E, seed])
        self.seed = seed
        self.n = int(r.choice([s for s in ([1, 64, 65, 257, 3000, 8193, 20_000] if not big else [300_000, 600_001]) if not max_n or s <= max_n]))
        self.depth = int(r.integers(1, 9))
        self.comps = [(f"C{i}", int(r.choice([1, 2, 4, 4, 8])), int(r.integers(1, 4))) for i in range(int(r.integers(2, 6)))]
        self.comps[0] = ("C0", 4, self.comps[0][2])                                     # add_u32 needs a 4-byte word somewhere
        self.no_rollback = int(r.integers(1, len(self.comps))) if r.random() < 0.2 else -1
        self.health = bool(r.random() < 0.5)
        self.health_mode = int(r.integers(0, 2))
        self.groups = []                                                                # (count, which components)
        left = self.n
        for g in range(int(r.integers(1, 4))):
            cnt = left if g == 2 else int(r.integers(0, left + 1))
            which = [i for i in range(len(self.comps)) if r.random() < 0.7] or [0]
            if cnt: self.groups.append((cnt, which))
            left -= cnt
        if left: self.groups.append((left, list(range(len(self.comps)))))
        self.cks = [(i, sorted(set(int(x) for x in r.integers(0, nw, int(r.integers(1, nw + 1)))))) for i, (_, wb, nw) in enumerate(self.comps)
                    if i != self.no_rollback and r.random() < 0.7]
        four = [(i, k) for i, (_, wb, nw) in enumerate(self.comps) if wb == 4 and i != self.no_rollback for k in range(nw)]
        self.adds = [four[int(r.integers(len(four)))] + (int(r.integers(1, 2**32)),) for _ in range(int(r.integers(1, 4)))]
        self.capacity = self.n + 8

    def describe(self):
        return (f"generic seed {self.seed}: n={self.n} depth={self.depth} comps={self.comps} no_rollback={self.no_rollback} groups={[(c, w) for c, w in self.groups]} "
                f"checksums={self.cks} add_u32={self.adds} health={self.health and ('immediate', 'rollback')[self.health_mode]}")

    def build(self, world):
        r = np.random.default_rng([7, self.seed])
        ids = [world.register_component(nm, wb, nw, rollback=(i != self.no_rollback)) for i, (nm, wb, nw) in enumerate(self.comps)]
        H = world.register_component("Health", 4, 1) if self.health else None
        for i, words in self.cks: world.checksum_component(ids[i], words)
        if self.health: world.checksum_component(H, [0])
        for i, k, inc in self.adds: world.add_system(bg.SYS_ADD_U32, comp=(ids[i],), word=(k,), iparam=(inc,))
        if self.health: world.add_system(bg.SYS_SAT_SUB_DESPAWN, comp=(H,), word=(0,), iparam=(1, self.health_mode))
        for cnt, which in self.groups:
            bundle = {ids[i]: [r.integers(0, 2 ** (8 * self.comps[i][1]), cnt, dtype=np.uint64).astype(_DT[self.comps[i][1]]) for _ in range(self.comps[i][2])] for i in which}
            if self.health: bundle[H] = [r.integers(1, 40, cnt).astype(np.uint32)]
            world.spawn(cnt, bundle)
        world.set_depth(self.depth)
        if hasattr(world, "set_synctest_check_distance"): world.set_synctest_check_distance(-1)
        return tuple(ids) + ((H,) if self.health else ())


class BoxScenario:
    """box_game (examples/box_game/box_game.rs:89-206): cubes steered by 1-4 players' input bytes, Transform and Velocity under
    rollback, Player outside it; arbitrary starting states so that every branch of move_cube_system is taken."""
    with_spawn, spawn_budget, spawn_fn = False, 0, None

    def __init__(self, seed, *, big=False, max_n=None):
        r = self.rng = np.random.default_rng([0xB0C5, seed])
        self.seed = seed
        self.n = int(r.choice([s for s in ([2, 64, 257, 5000, 8193] if not big else [300_000]) if not max_n or s <= max_n]))
        self.players = int(r.integers(1, 5))
        self.depth = int(r.integers(1, 9))
        self.cks = [bool(r.random() < 0.7), bool(r.random() < 0.7)]
        self.fps = int(r.choice([60, 60, 30, 144]))
        self.capacity = self.n + 8

    def describe(self):
        return f"box_game seed {self.seed}: n={self.n} players={self.players} depth={self.depth} checksums(T, V)={self.cks} fps={self.fps}"

    def build(self, world):
        from test_box_game import build_box
        ids, *_ = build_box(world, self.n, self.players, seed=self.seed, spread=True, checksums=[(c, [0, 1, 2]) for c in (0, 1) if self.cks[c]])
        world.set_frame_rate(self.fps)
        world.set_depth(self.depth)
        if hasattr(world, "set_synctest_check_distance"): world.set_synctest_check_distance(-1)
        return ids

    def inputs(self, frame):
        # a predicted input may differ from the one confirmed later: the same frame is not always advanced with the same bytes
        return tuple(int(x) for x in self.rng.integers(0, 16, self.players))


def _adv(sc, frame, spawn):
    if isinstance(sc, BoxScenario): return bg.AdvanceFrame(sc.inputs(frame))
    a = bg.AdvanceFrame((cm.INPUT_SPAWN if spawn else 0,))
    if spawn: a.spawn_vx, a.spawn_vy = sc.spawn_fn(frame)
    return a


def _gen_list(sc, st):
    """One request list; `st` = dict(F, ring, confirmed, spawns_left) is advanced as the list is built."""
    r, reqs = sc.rng, []
    ring = st["ring"]

    def spawn_now():
        if not sc.with_spawn or st["spawns_left"] <= 0 or r.random() > 0.3: return False
        st["spawns_left"] -= 1
        return True

    def adv():
        reqs.append(_adv(sc, st["F"], spawn_now())); st["F"] = w32(st["F"] + 1)

    def save():
        reqs.append(bg.SaveGameState(st["F"])); ring.push(st["F"]); ring.confirm(st["confirmed"])

    def loadable():
        return [f for f in ring.frames if f >= st["confirmed"]]

    def load(f):
        reqs.append(bg.LoadGameState(f)); ring.rollback(f); st["F"] = f

    shape = r.random()
    cand = loadable()
    if shape < 0.2 and cand:
        # SyncTest tick: roll back to the oldest loadable frame, resimulate with a Save per frame (schedule_systems.rs:85-118)
        f0, F = cand[-1], st["F"]                            # (the ring's frames run newest first)
        load(f0)
        for _ in range(ahead(F, f0)):
            adv(); save()
        adv()
    elif shape < 0.35 and cand:
        # P2P-shaped: [Load(F - r), Adv, (Save, Adv) x (r - 1)] + [Save(F), Adv]
        F = st["F"]; f0 = int(r.choice(cand))
        load(f0); adv()
        for _ in range(max(0, ahead(F, f0) - 1)):
            save(); adv()
        save(); adv()
    elif shape < 0.55 and cand and ring.depth >= 2:
        # speculative branches off one snapshot: every branch but the last leaves nothing behind (fanout.py's list shape)
        C = cand[0]; B = int(r.integers(2, 7)); D = int(r.integers(1, min(5, ring.depth)))      # (the branches' Saves must not evict C)
        for b in range(B):
            load(C)
            for i in range(D):
                adv(); save()
            adv()
    else:
        for _ in range(int(r.integers(1, 24))):
            x = r.random(); cand = loadable()
            if x < 0.45: adv()
            elif x < 0.8 or not cand: save()
            else: load(int(r.choice(cand)))
    return reqs


def _box_words(sc, r, wb, count):
    """Finite floats for Transform / Velocity words, a valid handle for Player (the game never holds a NaN or a handle without a player)."""
    if wb == 8: return r.integers(0, sc.players, count).astype(np.uint64)
    return cm.f32bits(r.uniform(-3, 3, count).astype(np.float32))


def _mutate(sc, st, A, B, ids):
    """Host-side edits between two lists, the same on both backends."""
    r = sc.rng
    x = r.random()
    if x < 0.35:
        c = w32(st["confirmed"] + int(r.integers(0, ahead(st["F"], st["confirmed"]) + 1)))
        st["confirmed"] = c
        st["ring"].confirm(c)
        A.set_confirmed(c); B.set_confirmed(c)
        return f"confirmed={c}"
    if x < 0.38:
        d = int(r.integers(1, 9))                          # sync_depth: MaxPredictionWindow changed (mod.rs:123-138, 263-273); the next push trims the ring
        st["ring"].depth = d
        A.set_depth(d); B.set_depth(d)
        return f"depth={d}"
    n = A.len
    if n == 0: return None
    alive = np.nonzero(A.alive_mask(n))[0]
    if x < 0.45 and alive.size:
        s = int(r.choice(alive)); A.despawn(s); B.despawn(s)
        return f"despawn({s})"
    if x < 0.5 and alive.size and hasattr(A, "despawn_rollback") and not isinstance(sc, Scenario):
        s = int(r.choice(alive)); A.despawn_rollback(s); B.despawn_rollback(s)
        return f"despawn_rollback({s})"
    V = ids[1] if isinstance(sc, Scenario) else ids[int(r.integers(len(ids)))]
    _, wb, nw = A._comps[V]
    if x < 0.6 and alive.size:
        s = int(r.choice(alive))
        if A.present_mask(V, n)[s]:
            A.remove_component(V, s); B.remove_component(V, s)
            return f"remove c{V}({s})"
        w = r.integers(0, 2 ** (8 * wb), nw, dtype=np.uint64).astype(_DT[wb])
        if isinstance(sc, Scenario): w = cm.f32bits(np.array([r.uniform(-50, 50), r.uniform(-50, 50), 0.0], dtype=np.float32))
        if isinstance(sc, BoxScenario): w = _box_words(sc, r, wb, nw)
        A.insert_component(V, s, w); B.insert_component(V, s, w)
        return f"insert c{V}({s})"
    if x < 0.68:
        first = int(r.integers(0, n)); cnt = int(r.integers(1, min(n - first, 700) + 1)); k = int(r.integers(nw))
        data = r.integers(0, 2 ** (8 * wb), cnt, dtype=np.uint64).astype(_DT[wb])
        if isinstance(sc, Scenario): data = cm.f32bits(r.uniform(-100, 100, cnt).astype(np.float32))
        if isinstance(sc, BoxScenario): data = _box_words(sc, r, wb, cnt)
        A.upload_word(V, k, first, data); B.upload_word(V, k, first, data)
        return f"upload c{V}w{k}[{first}:{first + cnt}]"
    return None


def _extra_state(w):
    out = {}
    if hasattr(w, "disabled_mask"):
        out["disabled"] = w.disabled_mask(); out["despawned_frames"] = np.where(out["disabled"], w.despawned_frames(), 0)
    return out


def run(seed, make_a, make_b, n_lists=30, big=False, state_every=6, generic=False, max_n=None, box=False, start_frame=None):
    sc = (BoxScenario if box else GenericScenario if generic else Scenario)(seed, big=big, max_n=max_n)
    A, B = make_a(sc), make_b(sc)
    log = [sc.describe()]
    try:
        ids = sc.build(A); idsb = sc.build(B)
        assert tuple(ids) == tuple(idsb)
        if start_frame is not None:                          # e.g. a session that has been running for 2^31 frames: the counters wrap
            A.set_frame(start_frame); B.set_frame(start_frame)
            A.set_confirmed(start_frame); B.set_confirmed(start_frame)
            log[0] += f" start_frame={start_frame}"
        st = {"F": A.frame, "ring": _Ring(sc.depth), "confirmed": 0 if start_frame is None else start_frame, "spawns_left": sc.spawn_budget}
        use_async = hasattr(B, "enqueue_requests")
        pending = []

        def drain():
            while pending:
                k0, want, ctx0 = pending.pop(0)
                got = B.collect_checksums()
                assert want == list(got), f"checksums of the enqueued list {k0} differ:\n{ctx0}\n{want}\n{got}"

        for k in range(n_lists):
            if sc.rng.random() < 0.6: drain()            # host edits with lists still in flight are part of the game (they are stream-ordered behind them)
            m = _mutate(sc, st, A, B, ids) if not pending or sc.rng.random() < 0.5 else None
            if m: log.append(m)
            reqs = _gen_list(sc, st)
            log.append(" ".join(("S%d" % q.frame) if isinstance(q, bg.SaveGameState) else ("L%d" % q.frame) if isinstance(q, bg.LoadGameState)
                                else ("A*" if q.inputs[0] and not box else "A") for q in reqs))
            ca = A.handle_requests(reqs)
            ctx = "\n".join(log[:1] + log[-6:])
            if use_async and sc.rng.random() < 0.4 and len(pending) < 3 and k != n_lists - 1:
                # up to three lists in flight on B (enqueue / collect: the host-side bookkeeping of a list happens at enqueue time)
                B.enqueue_requests(reqs); pending.append((k, list(ca), ctx))
                log[-1] += "   (enqueued)"
                continue
            drain()
            cb = B.handle_requests(reqs)
            assert list(ca) == list(cb), f"checksums differ after list {k}:\n{ctx}\n{ca}\n{cb}"
            assert (A.frame, A.len, A.snapshot_count()) == (B.frame, B.len, B.snapshot_count()) == (st["F"], A.len, A.snapshot_count()), \
                f"frame / len / snapshots differ after list {k}: {(A.frame, A.len, A.snapshot_count())} vs {(B.frame, B.len, B.snapshot_count())} (model frame {st['F']})\n{ctx}"
            if k % state_every == state_every - 1 or k == n_lists - 1:
                cm.assert_states_equal(cm.snapshot_state(A, ids), cm.snapshot_state(B, ids), f"after list {k}:\n{ctx}\n")
                cm.assert_states_equal(_extra_state(A), _extra_state(B), f"RollbackDespawned markers after list {k}:\n{ctx}\n")
                for f in (w32(st["F"] + d) for d in range(-10, 2)):
                    assert A.has_snapshot(f) == B.has_snapshot(f), f"has_snapshot({f}) differs after list {k}:\n{ctx}"
        return log
    finally:
        A.close(); B.close()
