"""The speculative fan-out through the C ABI (ggrs_hip_fanout_*: RCCL is called inside libggrs_hip.so) on the one GPU a gpurun box has: world size 1 over
the real RCCL, world sizes 2 / 3 / 8 over a shared-memory stand-in for librccl (tests/cpp/rccl_double.cpp; RCCL refuses two ranks per device).  Every
gathered Checksum(u128) table must equal the oracle's serial walk of every branch; adopted branch states must equal the oracle's straight-line simulation
with the true inputs (tests/test_fanout_gloo.py covers the world-size-2 control flow on CPU).  torch.distributed carries no data here."""
import os

import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm

pytestmark = pytest.mark.gpu


class _DeviceSpan:
    """A raw device allocation presented through __cuda_array_interface__ so torch can alias it without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def test_state_block_roundtrip_library_and_caller_arena():
    """ggrs_hip_state_bytes / _live_state_ptr / _adopt_live_state: a second world -- on a CALLER-provided arena (a torch tensor) -- adopts the first one's
    packed live block byte for byte (what a rank does with a block that reached it by other means than the library's own broadcast)."""
    import torch
    from bevy_ggrs_amd import _ffi
    dev = torch.device("cuda", 0)
    n, D = 700, 6
    cap = n + 2000
    w = bg.World(cap, max_depth=D + 2)
    ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(w, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(w, 3)
    fn = cm.frame_spawn_fn(50)
    for t in range(9):
        drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
    nbytes = int(_ffi.lib.ggrs_hip_arena_bytes(cap, D + 2, 3, 60))
    arena2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    w2 = bg.World(cap, max_depth=D + 2, arena_ptr=arena2.data_ptr(), arena_bytes=nbytes, stream=torch.cuda.current_stream(dev).cuda_stream)
    ids2 = cm.build_particles(w2, with_spawn=True, ttl_init=25)
    w2.spawn(0, {})
    nb = w.state_bytes()
    assert nb == w2.state_bytes()
    src = torch.as_tensor(_DeviceSpan(w.live_state_ptr(), nb), device=dev)
    assert w2.live_state_ptr() == arena2.data_ptr(), "the live block is the head of a caller-provided arena"
    arena2[:nb].copy_(src)
    torch.cuda.synchronize()
    w2.adopt_live_state()
    assert w2.len == w.len and w2.frame == w.frame
    cm.assert_states_equal(cm.snapshot_state(w2, ids2), cm.snapshot_state(w, ids), "adopted")
    assert w2.save() == w.save()


def _n_devices():
    from bevy_ggrs_amd import _ffi
    return int(_ffi.lib.ggrs_hip_device_count())


def _native_fanout_rank(rank, size, id_q, n, D, steps, bpr, q, scenario="walk", opt=None):
    """One rank of the C-ABI fan-out (ggrs_hip_fanout_*): RCCL is called inside libggrs_hip.so; torch is not involved.
    Rank r runs on HIP device r % devices (one rank per GPU wherever the box has them); rank 0 creates the ncclUniqueId and
    hands it to the others -- the host's only job in the real thing too."""
    try:
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout
        if rank == 0:
            id_bytes = RcclFanout.unique_id()
            for _ in range(size - 1): id_q.put(id_bytes)
        else:
            id_bytes = id_q.get(timeout=120)
        device = rank % max(1, _n_devices())

        class _Dist:                                          # SpeculativeFanout only asks for rank and size here
            def get_rank(self): return rank
            def get_world_size(self): return size

        cap = n + 100 * (steps + D + 2) * 2
        w = bg.World(cap, max_depth=D + 2, device=device)
        ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
        if rank == 0:
            vel, ttl = cm.synthetic_particles(n, ttl="despawn")
            cm.spawn_particles(w, ids, n, vel, ttl)
            for _ in range(3):
                w.advance((0,))
        else:
            w.spawn(0, {})                                    # seals the world: the layout is fixed, the state arrives by broadcast
        native = RcclFanout(w, rank, size, id_bytes)
        assert native.comm_info() == (rank, size, device)
        opt = opt or {}
        if scenario == "adopt":
            # confirm-speculate-adopt rounds under a scripted true-input sequence (tests/fanout_scenarios.py)
            import fanout_scenarios as fs
            fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=bpr, native=native, branch_input=fs.branch_input, confirmed_input=fs.true_input,
                                    spawn_fn=cm.frame_spawn_fn(fs.RATE), retain=opt.get("retain", "all"))
            seen = fs.run_adopt_session(fan, size * bpr, steps, broadcast_every=opt.get("broadcast_every", 0))
            # ... and the session goes on from the adopted world: two more plain steps, then one more Save of where it stands
            for _ in range(2):
                o = fan.step(); seen.append((o["confirmed_frame"], o["confirmed_checksum"]))
            fan.settle()
            seen.append((w.frame, w.save()))
            state = cm.snapshot_state(w, ids)
            native.close()
            q.put((rank, "ok", seen, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in state.items()}))
            return
        if scenario == "adopt_refused":
            # GGRS_ADOPT_BROADCAST of a frame the owner did NOT keep (newest-only retention, an older frame asked for): only the owner can know -- every rank must come
            # back with the error (nobody may be left waiting in the broadcast), nothing is adopted anywhere, and the next, valid adoption works
            import fanout_scenarios as fs
            from bevy_ggrs_amd import _ffi
            fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=bpr, native=native, branch_input=fs.branch_input, confirmed_input=fs.true_input,
                                    spawn_fn=cm.frame_spawn_fn(fs.RATE), retain="newest")
            fan.step(); fan.drain()
            C = fan.confirmed
            before = (w.frame, w.len)
            errs = []
            for owner_branch in (0, bpr):                                        # a branch of rank 0, then one of rank 1
                try:
                    native.adopt(owner_branch, C + 1, None, _ffi.ADOPT_BROADCAST)
                    errs.append(None)
                except bg.GgrsHipError as e:
                    errs.append((e.code, str(e)[:160]))
            assert (w.frame, w.len) == before, ((w.frame, w.len), before)
            native.adopt(bpr, C + D, None, _ffi.ADOPT_BROADCAST)                   # the newest frame of rank 1's first branch IS kept
            after = (w.frame, w.save())
            native.close()
            q.put((rank, "ok", errs, after))
            return
        fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=bpr, native=native, max_inflight=2,
                                branch_input=lambda b, f: cm.INPUT_SPAWN if b % 2 == 0 else 0,
                                confirmed_input=lambda f: cm.INPUT_SPAWN if f % 2 == 1 else 0,
                                spawn_fn=cm.frame_spawn_fn(50), compact=opt.get("compact", True), desync_detection_interval=opt.get("interval", 1))
        out = [fan.step() for _ in range(steps // 2)]
        fan.results.clear()
        for _ in range(steps - steps // 2):
            fan.step_pipelined()
        fan.drain()
        out += fan.results
        fan.settle()
        state = cm.snapshot_state(w, ids)
        native.close()
        q.put((rank, "ok", out, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in state.items()}))
    except Exception as e:                                    # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}", traceback.format_exc()))


def _run_native(size, n=700, D=4, steps=6, bpr=2, env=None, scenario="walk", opt=None):
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q, id_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_native_fanout_rank, args=(r, size, id_q, n, D, steps, bpr, q, scenario, opt)) for r in range(size)]
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})                              # spawned children inherit the parent's environment at start()
    try:
        for p in procs: p.start()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    res = {}
    try:
        for _ in range(size):
            r = q.get(timeout=240)
            res[r[0]] = r[1:]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive(): p.kill()
    return res


def test_native_fanout_world_size_1_matches_serial_reference():
    """ggrs_hip_fanout_{unique_id,init,sync_confirmed,step,collect,destroy}: ncclBroadcast of the live block and
    ncclAllGather of the checksums happen inside the library."""
    from test_fanout_gloo import _serial_reference
    n, D, steps, bpr = 700, 4, 6, 2
    res = _run_native(1, n, D, steps, bpr)
    assert res[0][0] == "ok", res[0]
    out, state = res[0][1], res[0][2]
    ref, ref_state = _serial_reference(n, D, bpr, steps)
    assert len(out) == len(ref) == steps
    for got, want in zip(out, ref):
        assert got["confirmed_frame"] == want["confirmed_frame"] and got["confirmed_checksum"] == want["confirmed_checksum"]
        assert got["branch_checksums"] == want["branch_checksums"]
    for k, v in ref_state.items():
        assert (np.asarray(state[k]) == np.asarray(v)).all(), k


@pytest.mark.parametrize("opt", [{"compact": False}, {"compact": True, "interval": 3}, {"compact": False, "interval": 3}])
def test_native_fanout_request_list_and_compact_step_agree(opt):
    """The compact step (ggrs_hip_fanout_step_branches: ONE launch for all branches, member records) and the request list it replaces (dead-snapshot
    elimination + batches of identical groups), with 1 and 3 steps per all-gather: the same tables as the serial reference."""
    from test_fanout_gloo import _serial_reference
    n, D, steps, bpr = 700, 4, 7, 3
    res = _run_native(1, n, D, steps, bpr, opt=opt)
    assert res[0][0] == "ok", res[0]
    ref, ref_state = _serial_reference(n, D, bpr, steps)
    assert len(res[0][1]) == steps
    for got, want in zip(res[0][1], ref):
        assert got["confirmed_checksum"] == want["confirmed_checksum"] and got["branch_checksums"] == want["branch_checksums"]
    for k, v in ref_state.items():
        assert (np.asarray(res[0][2][k]) == np.asarray(v)).all(), k


def _config5_rank(q):
    try:
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout, default_branch_input

        class _Dist:
            def get_rank(self): return 0
            def get_world_size(self): return 1
        n, D, steps, bpr, rate = 100_000, 8, 2, 256, 100
        cap = n + 2 * rate * (steps + D + 2) * 2
        w5 = bg.World(cap, max_depth=D + 2)
        ids5 = cm.build_particles(w5, with_spawn=True, ttl_init=300)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w5, ids5, n, vel, ttl)
        native = RcclFanout(w5, 0, 1, RcclFanout.unique_id())
        fan5 = SpeculativeFanout(w5, _Dist(), D, None, branches_per_rank=bpr, native=native, confirmed_input=lambda f: 0x13, spawn_fn=cm.frame_spawn_fn(rate))
        out5 = [fan5.step() for _ in range(steps)]
        fan5.settle()
        state = cm.snapshot_state(w5, ids5)
        native.close()
        q.put((0, "ok", out5, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in state.items()}))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put((0, "error", f"{type(e).__name__}: {e}", traceback.format_exc()))


def test_config5_256_diverging_branches_in_one_launch():
    """BASELINE config 5 on the one GPU of this box: 256 predicted-input branches (branch id = the input byte repeated every frame), 100 k
    entities, 8 frames each; inputs with INPUT_SPAWN set spawn 100 particles per frame, so 128 of the branches diverge from the others -- and all 256 ride in
    ONE launch per step (member records).  Every branch's checksums against the oracle's serial walk."""
    import multiprocessing as mp
    from bevy_ggrs_amd.fanout import default_branch_input
    from test_fanout_gloo import _serial_reference
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_config5_rank, args=(q,)); p.start()
    try: r = q.get(timeout=600)
    finally:
        p.join(timeout=30)
        if p.is_alive(): p.kill()
    assert r[1] == "ok", r
    out5, state = r[2], r[3]
    n, D, steps, bpr, rate = 100_000, 8, 2, 256, 100
    ref5, state5 = _serial_reference(n, D, bpr, steps, branch_input=default_branch_input, confirmed_input=lambda f: 0x13, ttl_init=300, rate=rate, warm=0)
    for got, want in zip(out5, ref5):
        assert got["confirmed_checksum"] == want["confirmed_checksum"]
        assert got["branch_checksums"] == want["branch_checksums"]
        assert len({v[0] for v in got["branch_checksums"].values()}) == 1          # one confirmed frame
        assert len({tuple(v) for v in got["branch_checksums"].values()}) == 2      # spawning vs non-spawning predictions
    for k, v in state5.items():
        assert (np.asarray(state[k]) == np.asarray(v)).all(), k


def _check_adopt(res, size, n, D, steps, bpr):
    import fanout_scenarios as fs
    assert all(r[0] == "ok" for r in res.values()), {k: v[:2] for k, v in res.items() if v[0] != "ok"}
    seen0 = res[0][1]
    last_frame = max(f for f, _ in seen0)
    cap = n + 100 * (steps * D + D + 8) * 2
    cs, _, _ = fs.straight_line_reference(n, last_frame - 3 + 1, cap)
    assert any(b - a > 1 for (a, _), (b, _) in zip(seen0, seen0[1:])), "no adoption ever jumped more than one frame: the scenario tests nothing"
    for r in range(size):
        seen, state = res[r][1], res[r][2]
        assert seen == seen0, f"rank {r} observed other frames / checksums than rank 0"
        for f, c in seen:
            assert c is None or cs[f] == c, (r, f, hex(c), hex(cs[f]))
        want = fs.state_at(n, state["frame"], cap)
        for k, v in want.items():
            assert (np.asarray(state[k]) == np.asarray(v)).all(), (r, k)


def test_adopt_world_size_1_over_rccl():
    """ggrs_hip_fanout_adopt at world size 1 (the real RCCL): every branch is this rank's -- adoption is a ring-slot swap + one LoadWorld, no re-simulation.
    After the scripted session the world equals the oracle's straight-line simulation of the true inputs, as does every checksum seen on the way."""
    n, D, steps, bpr = 700, 5, 8, 4
    res = _run_native(1, n, D, steps, bpr, scenario="adopt")
    _check_adopt(res, 1, n, D, steps, bpr)


def test_adopt_newest_only_world_size_1():
    """GGRS_BRANCH_RETAIN_NEWEST keeps only the last frame of every branch: adopting an earlier one is refused with GGRS_E_NO_SNAPSHOT, the newest one works."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_newest_rank, args=(q,)); p.start()
    try: r = q.get(timeout=300)
    finally:
        p.join(timeout=30)
        if p.is_alive(): p.kill()
    assert r[0] == "ok", r


def _newest_rank(q):
    try:
        import bevy_ggrs_amd as bg
        import common as cm
        import fanout_scenarios as fs
        from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout

        class _Dist:
            def get_rank(self): return 0
            def get_world_size(self): return 1
        n, D, bpr = 500, 4, 3
        w = bg.World(n + 4000, max_depth=D + 2)
        ids = fs.build_world(w, n, True)
        native = RcclFanout(w, 0, 1, RcclFanout.unique_id())
        always = lambda b, f: cm.INPUT_SPAWN                       # every branch predicts the key held, and so it is
        fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=bpr, native=native, branch_input=always, confirmed_input=lambda f: cm.INPUT_SPAWN,
                                spawn_fn=cm.frame_spawn_fn(fs.RATE), retain="newest")
        fan.step()
        try:
            fan.adopt(1, 2); raise AssertionError("adopting a frame that was not kept must fail")
        except bg.GgrsHipError as e:
            assert e.code == bg.GGRS_E_NO_SNAPSHOT and "retained" in str(e), e
        C = fan.confirmed
        fan.adopt(1, D)                                            # the newest frame: C + D (the state after the branch's last AdvanceFrame)
        assert w.frame == C + D and fan.confirmed == C + D
        got = cm.snapshot_state(w, ids)
        cs = w.save()
        # reference: the same inputs in a straight line
        from oracle.binding import OracleWorld
        o = OracleWorld(n + 4000, 4)
        oids = fs.build_world(o, n, True)
        fn = cm.frame_spawn_fn(fs.RATE)
        while o.frame < C + D:
            a = bg.AdvanceFrame((cm.INPUT_SPAWN,)); a.spawn_vx, a.spawn_vy = fn(o.frame)
            o.handle_requests([a])
        cm.assert_states_equal(got, cm.snapshot_state(o, oids), "newest adopted")
        o.set_depth(2)
        assert cs == o.save()
        native.close()
        q.put(("ok",))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put(("error", f"{type(e).__name__}: {e}", traceback.format_exc()))


@pytest.mark.parametrize("size,bpr,broadcast_every", [(2, 3, 0), (3, 2, 2), (8, 2, 3)])
def test_adopt_ranks_over_the_transport_double(size, bpr, broadcast_every):
    """Adoption ACROSS ranks (collectives over the shared-memory stand-in): the rank that ran the matching branch swaps its retained block into the ring, the
    others re-simulate with the confirmed inputs -- or, every `broadcast_every`-th round, receive the owner's block by one ncclBroadcast -- and every rank must
    end in the oracle's straight-line state with the same checksums on the way, whoever owned the adopted branches."""
    n, D, steps = 600, 5, 8
    res = _run_native(size, n, D, steps, bpr, env={"GGRS_RCCL_LIB": _double_lib()}, scenario="adopt", opt={"broadcast_every": broadcast_every})
    _check_adopt(res, size, n, D, steps, bpr)


def test_broadcast_adoption_is_refused_on_every_rank_when_the_owner_kept_no_such_frame():
    res = _run_native(2, 600, 4, 1, 2, env={"GGRS_RCCL_LIB": _double_lib()}, scenario="adopt_refused")
    assert all(res[r][0] == "ok" for r in (0, 1)), res
    e0, e1 = res[0][1], res[1][1]
    # branch 0 lives on rank 0: rank 0 reports what it lacks, rank 1 that rank 0 could not hand it over; and the other way round for branch 2
    assert e0[0] and e0[0][0] == bg.GGRS_E_NO_SNAPSHOT and "holds no retained state" in e0[0][1], e0
    assert e1[0] and e1[0][0] == bg.GGRS_E_NO_SNAPSHOT and "rank 0 cannot hand over" in e1[0][1], e1
    assert e1[1] and "holds no retained state" in e1[1][1] and e0[1] and "rank 1 cannot hand over" in e0[1][1], (e0, e1)
    assert res[0][2] == res[1][2], "the ranks differ after the valid adoption that followed"


def _double_lib():
    """tests/cpp/rccl_double.cpp built on demand: the shared-memory stand-in for librccl.so (same-box transport)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, out = os.path.join(root, "tests", "cpp", "rccl_double.cpp"), os.path.join(root, "tests", "cpp", "_build", "librccl_double.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    return out


@pytest.mark.parametrize("size", [2, 3])
def test_native_fanout_ranks_over_the_transport_double(size):
    """The rank != 0 half of ggrs_hip_fanout_* on a ONE-GPU box: RCCL refuses two ranks per device, so the collective library is
    swapped (GGRS_RCCL_LIB) for tests/cpp/rccl_double.cpp, which moves the same collectives through shared memory.  Everything
    else is the product: ncclBroadcast of rank 0's packed live block into the other ranks' HBM and adoption (len, frame, row
    versions), every rank's own branch lists, the all-gather of the Checksum(u128)s on the side stream, collect.  The gathered
    table must equal the serial reference of all size x bpr branches and every rank must end in the same confirmed state."""
    from test_fanout_gloo import _serial_reference
    n, D, steps, bpr = 700, 4, 6, 2
    res = _run_native(size, n, D, steps, bpr, env={"GGRS_RCCL_LIB": _double_lib()})
    assert all(r[0] == "ok" for r in res.values()), res
    ref, ref_state = _serial_reference(n, D, size * bpr, steps)
    for r in range(size):
        out, state = res[r][1], res[r][2]
        assert len(out) == steps
        for got, want in zip(out, ref):
            assert got["confirmed_checksum"] == want["confirmed_checksum"]
            assert got["branch_checksums"] == want["branch_checksums"]
        for k, v in ref_state.items():
            assert (np.asarray(state[k]) == np.asarray(v)).all(), (r, k)


def test_eight_ranks_config5_partition_over_the_transport_double():
    """BASELINE config 5's REAL partition -- 256 predicted-input branches over 8 ranks, 32 per rank -- run as eight processes on the one GPU of this box
    (collectives over the shared-memory stand-in; control plane: none needed, the unique id travels through a queue): every rank receives rank 0's
    confirmed snapshot, walks its 32 branches per step, and the gathered table of 256 branches must equal the oracle's serial walk on every rank
    (no N > 1 hardware exists for this build, so this is the multi-rank evidence)."""
    from test_fanout_gloo import _serial_reference
    size, n, D, steps, bpr = 8, 500, 4, 3, 32
    res = _run_native(size, n, D, steps, bpr, env={"GGRS_RCCL_LIB": _double_lib()})
    assert all(r[0] == "ok" for r in res.values()), {k: v[:2] for k, v in res.items() if v[0] != "ok"}
    ref, ref_state = _serial_reference(n, D, size * bpr, steps)
    for r in range(size):
        out, state = res[r][1], res[r][2]
        assert len(out) == steps
        for got, want in zip(out, ref):
            assert got["confirmed_checksum"] == want["confirmed_checksum"]
            assert len(got["branch_checksums"]) == 256 and got["branch_checksums"] == want["branch_checksums"]
        for k, v in ref_state.items():
            assert (np.asarray(state[k]) == np.asarray(v)).all(), (r, k)


def test_native_fanout_two_ranks_over_rccl():
    """World size 2 over the real RCCL, rank r on device r % devices: on a box with >= 2 GPUs this is the real thing; with both
    ranks on the one visible GPU, RCCL builds that refuse two ranks per device make this a skip, not a failure (the transport
    double above covers the library's side of it)."""
    from test_fanout_gloo import _serial_reference
    n, D, steps, bpr = 700, 4, 6, 2
    res = _run_native(2, n, D, steps, bpr)
    errs = [r for r in res.values() if r[0] != "ok"]
    if errs:
        msg = " | ".join(str(e[1]) for e in errs)
        if "ncclCommInitRank" in msg or "uplicate" in msg or "invalid usage" in msg.lower():
            pytest.skip(f"this RCCL refuses two ranks on one GPU: {msg[:200]}")
        raise AssertionError(errs)
    ref, _ = _serial_reference(n, D, 2 * bpr, steps)
    for r in (0, 1):
        out = res[r][1]
        assert len(out) == steps
        for got, want in zip(out, ref):
            assert got["confirmed_checksum"] == want["confirmed_checksum"]
            assert got["branch_checksums"] == want["branch_checksums"]


def _bench_line(args, env=None, timeout=900):
    """Run bench.py as the driver does (a fresh process) and return its ONE JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env={**os.environ, **(env or {})}, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]                   # stdout is the line and NOTHING else (no library banner before or after it)
    assert r.returncode == 0 and len(lines) == 1 and lines[0].startswith("{"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


def test_bench_two_ranks_line_carries_parity_cpu_baseline_and_roofline():
    """`bench.py --gpus 2` end to end on the one GPU of this box: bench.py starts its two ranks itself, both on device 0
    (--oversubscribe; control plane gloo, the library's collectives over the shared-memory stand-in for RCCL), times its steps, and rank 0
    prints the line a SCALE run would record: n_gpus from ncclCommCount, the in-run parity of the gathered checksum table against the
    oracle's serial walk of every branch of both ranks, the CPU baseline and the roofline of rank 0's kernel."""
    line = _bench_line(["--gpus", "2", "--oversubscribe", "--steps", "6", "--warmup", "2", "--preheat-ms", "20", "--entities", "300000", "--cpu-ticks", "1"],
                       env={"GGRS_RCCL_LIB": _double_lib()})
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak"
    par = line["parity"]
    assert par["equal"] is True and par["checked_branches"] == 2 and par["checked_steps"] >= 1 and par["checked_saves"] == par["checked_steps"] * 2 * 8, par
    cb = line["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] == 1 and cb["flat_soa_port"]["value"] > 0, cb
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["launches_timed"] > 0 and roof["algorithmic_bytes_per_launch"] > 0, roof
    assert line["value"] > 0 and "2 ranks" in line["config"]["parallelism"]


def test_bench_config5_fanout_line_on_one_gpu():
    """BASELINE config 5 through bench.py at world size 1 (the real RCCL): 256 branches, the parity gate walks every one of them on the
    oracle, and the line carries the integer-multiply roofline next to the HBM one."""
    line = _bench_line(["--config", "5", "--steps", "4", "--warmup", "2", "--preheat-ms", "20", "--cpu-ticks", "1"])
    assert line["n_gpus"] == 1 and line["parity"]["equal"] is True and line["parity"]["checked_branches"] == 256, line.get("parity")
    assert line["cpu_baseline"]["value"] > 0
    assert line["roofline_alu"]["achieved"] > 0


def test_bench_two_ranks_under_torch_distributed_run():
    """The driver's launch form for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...` -- with N = 2 on the one GPU of this box: the ranks read RANK / LOCAL_RANK / WORLD_SIZE from the launcher's
    environment (no re-spawn), agree on ONE pre-heat step count, and rank 0 prints the one JSON line."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "12", "--warmup", "3", "--preheat-ms", "30", "--entities", "300000", "--cpu-ticks", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**os.environ, "GGRS_RCCL_LIB": _double_lib()}, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["parity"]["equal"] is True and line["cpu_baseline"]["value"] > 0
    assert line["preheat"]["same_step_count_on_every_rank"] is True and line["preheat"]["ticks"] >= 20


def _skewed_rank(rank, size, id_q, q, mode="frame"):
    """Rank 1 advances its confirmed frame once on its own before the common step: the ranks' lists then start at different frames."""
    try:
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout
        if rank == 0:
            id_bytes = RcclFanout.unique_id()
            for _ in range(size - 1): id_q.put(id_bytes)
        else:
            id_bytes = id_q.get(timeout=120)

        class _Dist:
            def get_rank(self): return rank
            def get_world_size(self): return size
        n, D = 900, 4
        w = bg.World(n + 64, max_depth=D + 2, device=0)
        ids = cm.build_particles(w)
        if rank == 0:
            vel, ttl = cm.synthetic_particles(n, ttl="throughput")
            cm.spawn_particles(w, ids, n, vel, ttl)
        else:
            w.spawn(0, {})
        native = RcclFanout(w, rank, size, id_bytes)
        if mode == "shape":                                           # rank 1 walks three branches per step, rank 0 two: the lists hold different numbers of Saves
            fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=3 if rank == 1 else 2, native=native)
            fan.sync_confirmed(0)
            try:
                fan.step(); verdict = "no error"
            except bg.GgrsHipError as e:
                verdict = f"GgrsHipError {e.code}: {e}"
            native.close()
            q.put((rank, "ok", True, verdict))
            return
        fan = SpeculativeFanout(w, _Dist(), D, None, branches_per_rank=2, native=native)
        fan.sync_confirmed(0)
        first = fan.step()                                            # in step: fine
        if mode == "refuse":
            # ... then rank 1 hands the library a list of another shape (one SaveGameState where the agreed step has 1 + 2 x D): refused on rank 1 at once; rank 0's step
            # goes through, and its collect must come back with "rank 1 refused" -- not wait in the all-gather for a rank that has stopped calling
            try:
                if rank == 1:
                    C = fan.confirmed
                    native.step([bg.LoadGameState(C), bg.AdvanceFrame((0,)), bg.SaveGameState(C + 1)])
                else:
                    fan.step()
                verdict = "no error"
            except bg.GgrsHipError as e:
                verdict = f"GgrsHipError {e.code}: {e}"
            native.close()
            q.put((rank, "ok", first is not None, verdict))
            return
        if rank == 1:                                                 # ... then rank 1 runs ahead by one confirmed frame
            C = fan.confirmed
            w.set_confirmed(C)
            w.handle_requests([bg.LoadGameState(C), bg.AdvanceFrame((0,)), bg.SaveGameState(C + 1)])
            fan.confirmed = C + 1
        try:
            fan.step()
            verdict = "no error"
        except bg.GgrsHipError as e:
            verdict = f"GgrsHipError {e.code}: {e}"
        except Exception as e:                                        # noqa: BLE001
            verdict = f"{type(e).__name__}: {e}"
        native.close()
        q.put((rank, "ok", first is not None, verdict))
    except Exception as e:                                            # noqa: BLE001
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}", traceback.format_exc()))


@pytest.mark.parametrize("mode", ["frame", "shape", "refuse"])
def test_ranks_out_of_step_are_refused_by_the_library(mode):
    """Collectives pair up by order: a rank that ran ahead would gather ANOTHER frame's checksums into the table (bench.py's clock-based
    pre-heat did, round 4).  Every step carries a tag {frame of its first request, saves} behind its checksums through the all-gather,
    and ggrs_hip_fanout_collect refuses a table whose ranks disagree -- on every rank, naming both frames ("frame").
    "shape": ranks whose lists hold different numbers of SaveGameState requests used to hand RCCL mismatched counts (undefined: DESIGN 9.8 of
    round 4); the first step now exchanges {interval, saves per step} in one small all-gather of its own and every rank refuses.
    "refuse": a rank whose step is refused AFTER the agreement (a list of another shape, a spawn beyond its capacity ..) still takes part in the group's all-gather, with
    a tag that says so: the other ranks' collect names it instead of waiting for ever."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, id_q = ctx.Queue(), ctx.Queue()
    old = os.environ.get("GGRS_RCCL_LIB")
    os.environ["GGRS_RCCL_LIB"] = _double_lib()
    try:
        procs = [ctx.Process(target=_skewed_rank, args=(r, 2, id_q, q, mode)) for r in range(2)]
        for p in procs: p.start()
    finally:
        if old is None: os.environ.pop("GGRS_RCCL_LIB", None)
        else: os.environ["GGRS_RCCL_LIB"] = old
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=240); res[r[0]] = r[1:]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive(): p.kill()
    for r in (0, 1):
        assert res[r][0] == "ok" and res[r][1] is True, res[r]
        if mode == "frame": assert "GgrsHipError" in res[r][2] and "out of step" in res[r][2] and "frame" in res[r][2], res[r]
        elif mode == "shape": assert "GgrsHipError" in res[r][2] and "disagree on the shape" in res[r][2], res[r]
        else: assert "GgrsHipError" in res[r][2] and ("every step of this fan-out holds" if r == 1 else "rank 1 refused step 0") in res[r][2], res[r]


SHOT_SPAWN = """
struct Shot { float x, y, vx, vy; };
__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload) {      // commands.spawn((Transform, Velocity, Ttl, Rollback))
    const Shot* s = reinterpret_cast<const Shot*>(payload);                // this entity's record (payload_stride = 16)
    e.f32(0) = s->x; e.f32(1) = s->y; e.f32(2) = s->vx; e.f32(3) = s->vy;
    e.u64(4) = (ggrs_u64)f.iparam[0] + (k & 3ull) + (ggrs_u64)(f.frame & 1);
}
"""
FIRE = 0x10


def _shot_world(w, ttl):
    """particles.rs' update + TTL-despawn systems (built in) and a USER-WRITTEN spawn system that builds each new entity from its own 16-byte record."""
    import struct
    import bevy_ggrs_amd as bg
    import common as cm
    from oracle.binding import OracleWorld
    T, V, L = cm.build_particles(w)
    binds = [(T, 0), (T, 1), (V, 0), (V, 1), (L, 0)]
    if isinstance(w, OracleWorld):
        def spawn(words, slot, k, f, payload):
            x, y, vx, vy = struct.unpack("<4I", bytes(payload[:16]))
            return [x, y, vx, vy, int(f.iparam[0]) + (k & 3) + (f.frame & 1)]
        w.add_spawn_system(spawn, bundle=(T, V, L), bindings=binds, payload_stride=16, iparam=(ttl,))
    else:
        w.add_spawn_system(SHOT_SPAWN, bundle=(T, V, L), bindings=binds, payload_stride=16, iparam=(ttl,), name="fire")
    return T, V, L


def _shots(frame, fire):
    import numpy as np
    n = 5 if fire else 0
    return n, np.random.default_rng([11, frame]).uniform(-50, 50, (n, 4)).astype(np.float32)


def _branch_step_rank(q, kind="shots"):
    """ggrs_hip_fanout_step_branches through ctypes, no Python driver in between: a world with a USER-WRITTEN spawn system (a 16-byte payload record per new entity),
    5 branches x 4 frames that fire in different frames, one shared spawn table; adoption of a branch that spawned; then the refusals."""
    try:
        import ctypes as C
        import numpy as np
        import bevy_ggrs_amd as bg
        import common as cm
        from bevy_ggrs_amd import _ffi
        from bevy_ggrs_amd.fanout import RcclFanout
        from oracle.binding import FLAT, OracleWorld
        n, B, T = 3000, 5, 4
        fires = np.array([[1, 0, 1, 0], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=bool)     # [branch][frame]: the player holds FIRE
        out = {}
        for name, w in (("gpu", bg.World(n + 200, max_depth=6)), ("oracle", OracleWorld(n + 200, 6, FLAT))):
            if kind == "shots":
                ids = _shot_world(w, ttl=9)
                vel, ttl = cm.synthetic_particles(n, ttl="despawn")
                cm.spawn_particles(w, ids, n, vel, ttl)
            else:
                # the bullets world of test_gpu_round5: a USER-WRITTEN system (moves, counts a life down, despawns) next to the user-written spawner.  Its source names
                # neither despawn_rollback() nor `kill`, so it can leave no RollbackDespawned marker and the world is open to branch steps
                import test_gpu_round5 as r5
                P, V, L, K = r5._bullet_world(w, life=9)
                rng = np.random.default_rng(3)
                pos = rng.uniform(-10, 10, (n, 2)).astype(np.float32); vel = rng.uniform(-5, 5, (n, 2)).astype(np.float32)
                w.spawn(n, {P: [cm.f32bits(pos[:, 0]), cm.f32bits(pos[:, 1])], V: [cm.f32bits(vel[:, 0]), cm.f32bits(vel[:, 1])], L: [(2 + np.arange(n) % 23).astype(np.uint32)], K: [(np.arange(n) % 5).astype(np.uint8)]})
            w.set_depth(6)
            w.handle_requests([bg.AdvanceFrame((0,)), bg.AdvanceFrame((0,))])
            w.set_confirmed(w.frame)
            F = w.frame
            shots = {i: _shots(F + i, True) for i in range(T)}                                                     # the payload of a firing frame: a function of the frame
            if name == "oracle":
                cs = w.handle_requests([bg.SaveGameState(F)])
                for b in range(B):
                    reqs = [bg.LoadGameState(F)]
                    for i in range(T):
                        a = bg.AdvanceFrame((FIRE if fires[b, i] else 0,))
                        if fires[b, i]: a.spawn_count, a.spawn_payload = shots[i][0], shots[i][1]
                        reqs += [a, bg.SaveGameState(F + 1 + i)]
                    cs += w.handle_requests(reqs)
                out[name] = cs
                continue
            native = RcclFanout(w, 0, 1, RcclFanout.unique_id())
            lib, fp = _ffi.lib, native._p
            pre, keep, _ = w.build_requests([bg.SaveGameState(F)])
            table = (_ffi.BranchSpawn * T)()
            pay = []
            for i in range(T):
                cnt, rec = shots[i]
                rec = np.ascontiguousarray(rec); pay.append(rec)
                table[i].count, table[i].payload, table[i].payload_bytes = cnt, rec.ctypes.data, rec.nbytes
            inputs = np.zeros((B, T, 1), dtype=np.uint8); inputs[:, :, 0] = np.where(fires, FIRE, 0)
            sel = np.where(fires, np.arange(1, T + 1, dtype=np.uint16)[None, :], 0).astype(np.uint16)
            bs = _ffi.BranchStep()
            bs.prefix, bs.n_prefix, bs.n_branches, bs.n_frames, bs.n_inputs, bs.flags = pre, 1, B, T, 1, _ffi.BRANCH_SAVE_LAST | _ffi.BRANCH_RETAIN_ALL
            bs.inputs, bs.spawn_table, bs.n_spawn_table, bs.spawn_sel = inputs.ctypes.data, table, T, sel.ctypes.data
            ns = C.c_uint32(0)
            rc = lib.ggrs_hip_fanout_step_branches(fp, C.byref(bs), C.byref(ns))
            assert rc == 0, lib.ggrs_hip_fanout_last_error(fp)
            assert ns.value == 1 + B * T, ns.value
            table_out = native.collect()
            out[name] = [int(p[0]) | (int(p[1]) << 64) for p in table_out.reshape(-1, 2)]
            assert w.frame == F and w.len == n                                                                    # speculation: the world did not move
            # adopt the branch that fired in every frame, at its last frame; the world then holds its new entities
            native.adopt(2, F + T)
            assert w.frame == F + T and w.len == n + sum(shots[i][0] for i in range(T)), (w.frame, w.len)
            out["adopted_save"] = w.save()
            # ---- refusals: nothing of these may touch the world
            pre2, keep2, _ = w.build_requests([bg.SaveGameState(w.frame)])                                       # the same shape as the agreed one: 1 + B x T SaveGameStates
            def refuse(**kw):
                b2 = _ffi.BranchStep(); C.memmove(C.byref(b2), C.byref(bs), C.sizeof(bs))
                b2.prefix = pre2
                for k, v in kw.items(): setattr(b2, k, v)
                rc = lib.ggrs_hip_fanout_step_branches(fp, C.byref(b2), None)
                return rc, (lib.ggrs_hip_fanout_last_error(fp) or b"").decode()
            bad_sel = sel.copy(); bad_sel[1, 1] = T + 3
            errs = [refuse(flags=64 | _ffi.BRANCH_SAVE_LAST | _ffi.BRANCH_RETAIN_ALL), refuse(n_frames=40), refuse(spawn_sel=bad_sel.ctypes.data), refuse(n_branches=0)]
            out["errs"] = [(rc, msg[:200]) for rc, msg in errs]
            out["after_refusals"] = (w.frame, w.len, w.save())
            native.close()
        q.put(("ok", out))
    except Exception as e:                                    # noqa: BLE001
        import traceback
        q.put(("error", f"{type(e).__name__}: {e}", traceback.format_exc()))


@pytest.mark.parametrize("kind", ["shots", "bullets"])
def test_branch_step_with_a_user_written_spawner_through_the_c_abi(kind):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_branch_step_rank, args=(q, kind)); p.start()
    try: r = q.get(timeout=600)
    finally:
        p.join(timeout=30)
        if p.is_alive(): p.kill()
    assert r[0] == "ok", r
    out = r[1]
    assert out["gpu"] == out["oracle"], "the branch step's checksums differ from the oracle's request lists"
    assert out["adopted_save"] == out["oracle"][1 + 2 * 4 + 3]                                   # branch 2's last SaveGameState
    assert len(set(out["gpu"][1 + b * 4 + 3] for b in range(5))) == 5                            # five different futures
    e = out["errs"]
    assert all(rc == bg.GGRS_E_INVALID for rc, _ in e), e
    assert "flags" in e[0][1] and "spawn_sel" in e[2][1] and "branches" in e[3][1], e
    assert out["after_refusals"][2] == out["adopted_save"], "a refused branch step moved the world"
