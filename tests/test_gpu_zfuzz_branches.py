"""ggrs_hip_fanout_step_branches / ggrs_hip_fanout_adopt, FUZZED against the oracle's request lists (each branch re-simulates from the confirmed frame on its
own copy; the matching branch's state is adopted).  Per seed: a world of random size / schema / spawn system, three consecutive branch steps with random branch counts'
inputs, spawn selections and retention flags, every Checksum(u128) of the gathered table compared with the oracle walking the same branches as request lists; then a
random branch is adopted at a random retained frame and the WHOLE live state (every column, mask and len) is compared with the oracle's replay of that branch.  One
process runs all seeds (one RCCL communicator per world, world size 1: the collective is real, the transport trivial)."""
import multiprocessing as mp

import pytest

pytestmark = pytest.mark.gpu

SEEDS = list(range(7100, 7136))


def _fuzz_rank(q, seeds):
    try:
        import ctypes as C
        import numpy as np
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd import _ffi
        from bevy_ggrs_amd.fanout import RcclFanout
        from oracle.binding import FLAT, OracleWorld
        lib = _ffi.lib
        report = []
        for seed in seeds:
            rng = np.random.default_rng(seed)
            n = int(rng.choice([300, 1500, 6000, 40_000]))
            depth = int(rng.integers(3, 9))
            T = int(rng.integers(1, depth + 1))                     # frames per branch (T SaveGameStates with SAVE_LAST, T - 1 without)
            B = int(rng.choice([1, 2, 5, 17, 40]))
            schema = str(rng.choice(["headline", "full", "allhot"]))
            with_spawn = bool(rng.integers(0, 2))
            save_last = bool(rng.integers(0, 2)) or T == 1
            retain = int(rng.choice([0, _ffi.BRANCH_RETAIN_NEWEST, _ffi.BRANCH_RETAIN_ALL]))
            flags = (_ffi.BRANCH_SAVE_LAST if save_last else 0) | retain
            n_spawn = int(rng.integers(1, 40))
            steps = 3
            cap = n + n_spawn * (T + 2) * (steps + 2) + 64
            what = dict(seed=seed, n=n, depth=depth, T=T, B=B, schema=schema, with_spawn=with_spawn, flags=flags, value_tags=bool(seed % 2))

            def payload(frame):                                      # a pure function of the frame: every branch that spawns in it draws the same entities
                r = np.random.default_rng([seed, frame])
                return r.uniform(-200, 200, n_spawn).astype(np.float32), r.uniform(-200, 200, n_spawn).astype(np.float32)

            ttl_init = int(rng.integers(2, 30))
            gw, ow = bg.World(cap, max_depth=depth + 2), OracleWorld(cap, depth + 2, FLAT)
            if seed % 2:                                             # value tags forced on (by default only worlds whose Save is bound by bytes keep them), lazy live block forced
                assert lib.ggrs_dbg_set_value_tags(gw._p, 1) == 0 and lib.ggrs_dbg_set_lazy_live(gw._p, 2 if seed % 4 == 1 else 1) == 0
            ids = None
            for w in (gw, ow):
                ids = cm.build_particles(w, with_spawn=with_spawn, ttl_init=ttl_init, schema=schema)
                vel, ttl = cm.synthetic_particles(n, ttl="despawn", seed=seed)
                cm.spawn_particles(w, ids, n, vel, ttl)
                w.set_depth(depth + 1)
                w.handle_requests([bg.AdvanceFrame((0,)), bg.AdvanceFrame((0,))])
            native = RcclFanout(gw, 0, 1, RcclFanout.unique_id())
            fp = native._p
            last = None
            for step in range(steps):
                Cf = gw.frame
                assert ow.frame == Cf
                for w in (gw, ow): w.set_confirmed(Cf)
                c_in = int(rng.choice([0, cm.INPUT_SPAWN])) if with_spawn else 0
                if step == 0:
                    prefix = [bg.SaveGameState(Cf)]
                else:
                    a = bg.AdvanceFrame((c_in,))
                    if c_in & cm.INPUT_SPAWN: a.spawn_vx, a.spawn_vy = payload(Cf)
                    prefix = [bg.LoadGameState(Cf), a, bg.SaveGameState(Cf + 1)]
                F = Cf if step == 0 else Cf + 1                      # the frame the branches start from
                pred = rng.choice([0, cm.INPUT_SPAWN] if with_spawn else [0, 1, 2], size=(B, T)).astype(np.uint8)
                # ---- oracle: the prefix, then every branch as its own request list
                want = list(ow.handle_requests(prefix))
                for b in range(B):
                    reqs = [bg.LoadGameState(F)]
                    for i in range(T):
                        a = bg.AdvanceFrame((int(pred[b, i]),))
                        if with_spawn and (pred[b, i] & cm.INPUT_SPAWN): a.spawn_vx, a.spawn_vy = payload(F + i)
                        reqs.append(a)
                        if i < T - 1 or save_last: reqs.append(bg.SaveGameState(F + 1 + i))
                    want += list(ow.handle_requests(reqs))
                ow.handle_requests([bg.LoadGameState(F)])            # the branch step leaves the world at the confirmed frame
                # ---- library: one call
                pre, keep, _ = gw.build_requests(prefix)
                table = (_ffi.BranchSpawn * T)()
                pays = []
                for i in range(T):
                    vx, vy = payload(F + i); pays.append((vx, vy))
                    table[i].count, table[i].vx, table[i].vy = n_spawn, vx.ctypes.data, vy.ctypes.data
                inputs = np.ascontiguousarray(pred.reshape(B, T, 1))
                sel = np.where((pred & cm.INPUT_SPAWN) != 0, np.arange(1, T + 1, dtype=np.uint16)[None, :], 0).astype(np.uint16) if with_spawn else np.zeros((B, T), dtype=np.uint16)
                bs = _ffi.BranchStep()
                bs.prefix, bs.n_prefix, bs.n_branches, bs.n_frames, bs.n_inputs, bs.flags = pre, len(prefix), B, T, 1, flags
                bs.inputs = inputs.ctypes.data
                if with_spawn: bs.spawn_table, bs.n_spawn_table, bs.spawn_sel = table, T, sel.ctypes.data
                ns = C.c_uint32(0)
                rc = lib.ggrs_hip_fanout_step_branches(fp, C.byref(bs), C.byref(ns))
                assert rc == 0, (what, step, lib.ggrs_hip_fanout_last_error(fp))
                got = [int(p[0]) | (int(p[1]) << 64) for p in native.collect().reshape(-1, 2)]
                assert ns.value == len(want) == len(got), (what, step, ns.value, len(want), len(got))
                bad = [k for k in range(len(want)) if want[k] != got[k]]
                assert not bad, (what, step, "first differing Checksum(u128) at", bad[0], "of", len(want))
                assert gw.frame == F and gw.len == ow.len, (what, step, gw.frame, F, gw.len, ow.len)
                last = (F, pred)
            # ---- adoption of a random branch at a random retained frame
            F, pred = last
            adopted = None
            if retain:
                b = int(rng.integers(0, B))
                if retain == _ffi.BRANCH_RETAIN_ALL: k = int(rng.integers(1, T + 1))
                else: k = T
                native.adopt(b, F + k)
                reqs = [bg.LoadGameState(F)]
                for i in range(k):
                    a = bg.AdvanceFrame((int(pred[b, i]),))
                    if with_spawn and (pred[b, i] & cm.INPUT_SPAWN): a.spawn_vx, a.spawn_vy = payload(F + i)
                    reqs.append(a)
                ow.handle_requests(reqs)
                assert gw.frame == ow.frame == F + k and gw.len == ow.len, (what, "adopt", b, k, gw.frame, ow.frame, gw.len, ow.len)
                cm.assert_states_equal(cm.snapshot_state(gw, ids), cm.snapshot_state(ow, ids), f"{what} adopted branch {b} at +{k}")
                assert gw.save() == ow.save(), (what, "adopt: SaveGameState after adoption")
                # and the session goes on from there: one more ordinary tick on both
                for w in (gw, ow): w.set_confirmed(w.frame)
                t1 = [bg.SaveGameState(F + k), bg.AdvanceFrame((0,)), bg.SaveGameState(F + k + 1)]
                assert list(gw.handle_requests(t1)) == list(ow.handle_requests(t1)), (what, "the tick after adoption")
                adopted = (b, k)
            native.close()
            what["adopted"] = adopted
            report.append(what)
        q.put(("ok", report))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put(("error", f"{type(e).__name__}: {e}", traceback.format_exc()))


@pytest.mark.parametrize("chunk", [0, 1, 2])
def test_branch_steps_and_adoption_fuzzed_against_the_oracle(chunk):
    seeds = SEEDS[chunk::3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_fuzz_rank, args=(q, seeds)); p.start()
    try: r = q.get(timeout=900)
    finally:
        p.join(timeout=30)
        if p.is_alive(): p.kill()
    assert r[0] == "ok", r
    assert len(r[1]) == len(seeds)
