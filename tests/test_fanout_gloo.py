"""N > 1 path on CPU: SpeculativeFanout's control flow (branch assignment, request lists,
replicated confirmed advance, the single all-gather, desync detection) with world_size 2 over
gloo.  The worlds here are oracle worlds (test infrastructure); the packed-state broadcast is
replaced by a host-side exchange with the same contract.  The HIP/RCCL exchange itself is
covered by tests/test_gpu_zfanout.py (world_size 1 over nccl on the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class HostExchange:
    """Same contract as bevy_ggrs_amd.fanout.HipStateExchange, over host arrays + gloo."""

    def __init__(self, world, ids):
        self.world, self.ids = world, ids

    def broadcast(self, d, src):
        w = self.world
        box = [None]
        if d.get_rank() == src:
            n = w.len
            st = {"len": n, "frame": w.frame, "alive": w.alive_mask(n)}
            for cid in self.ids:
                _, wb, nw = w._comps[cid]
                st[f"p{cid}"] = w.present_mask(cid, n)
                for k in range(nw):
                    st[f"c{cid}w{k}"] = w.download_word(cid, k, 0, n)
            box = [st]
        d.broadcast_object_list(box, src=src)
        st = box[0]
        if d.get_rank() != src:
            n = st["len"]
            assert w.len <= n
            if w.len < n:
                w.spawn(n - w.len, {cid: None for cid in self.ids})
            for cid in self.ids:
                _, wb, nw = w._comps[cid]
                for k in range(nw):
                    w.upload_word(cid, k, 0, st[f"c{cid}w{k}"])
                for s in np.nonzero(~st[f"p{cid}"])[0]:
                    w.remove_component(cid, int(s))
            for s in np.nonzero(~st["alive"])[0]:
                w.despawn(int(s))
            w.set_frame(st["frame"])

    def all_gather_u64(self, d, values):
        t = torch.from_numpy(values.view(np.int64).copy())
        out = [torch.empty_like(t) for _ in range(d.get_world_size())]
        d.all_gather(out, t)
        return torch.stack(out).numpy().view(np.uint64)


def _queued_oracle_world():
    """An oracle world with the library's enqueue / collect entry points (run at once, results queued): what SpeculativeFanout's
    pipelined path needs, so that its pre-marshalled request template, its steps in flight and its one all-gather per
    `desync_detection_interval` steps run on the CPU too."""
    import ctypes as C
    from bevy_ggrs_amd import _ffi
    from oracle.binding import OracleWorld, lib

    class QueuedOracleWorld(OracleWorld):
        def __init__(self, *a, **k):
            super().__init__(*a, **k); self._queue = []

        def enqueue_requests_raw(self, arr, n):
            n_save = sum(1 for i in range(n) if arr[i].kind == _ffi.REQ_SAVE)
            out = (C.c_uint64 * max(2, 2 * n_save))()
            self.handle_requests_raw(arr, n, out)
            self._queue.append((out, n_save))

        def collect_checksums_raw(self, out, max_saves):
            src, n_save = self._queue.pop(0)
            assert n_save <= max_saves
            C.memmove(out, src, 16 * n_save)

        def enqueue_requests(self, requests):
            arr, keep, n_save = self.build_requests(requests)
            self.enqueue_requests_raw(arr, len(requests))
            return n_save

        def collect_checksums(self, max_saves=256):
            src, n_save = self._queue.pop(0)
            return [int(src[2 * i]) | (int(src[2 * i + 1]) << 64) for i in range(n_save)]

    return QueuedOracleWorld


def _worker(rank, world_size, port, n, D, bpr, steps, corrupt, q, pipelined=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        import common as cm
        from bevy_ggrs_amd.fanout import DesyncDetected, SpeculativeFanout
        from bevy_ggrs_amd import SaveGameState as bg_SaveGameState
        from oracle.binding import OracleWorld
        cap = n + 100 * (steps + D + 2) * 2
        w = (_queued_oracle_world() if pipelined else OracleWorld)(cap, D + 1)
        ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
        if rank == 0:                              # only the root owns the confirmed world
            vel, ttl = cm.synthetic_particles(n, ttl="despawn")
            cm.spawn_particles(w, ids, n, vel, ttl)
            for _ in range(3):
                w.advance((0,))
        fn = cm.frame_spawn_fn(50)
        # branches: even ids press INPUT_SPAWN; the confirmed input presses on odd frames
        fan = SpeculativeFanout(w, dist, D, HostExchange(w, ids), branches_per_rank=bpr,
                                branch_input=lambda b, f: cm.INPUT_SPAWN if b % 2 == 0 else 0,
                                confirmed_input=lambda f: cm.INPUT_SPAWN if f % 2 == 1 else 0,
                                spawn_fn=fn, max_inflight=2 if pipelined else 1, desync_detection_interval=pipelined or 1)
        out = []
        err = None
        for s in range(steps):
            if corrupt and s == corrupt and rank == 1:
                # replica drifts: back on the confirmed frame a live particle's velocity changes and the confirmed
                # snapshot is re-saved
                fan.settle()
                w.upload_word(ids[1], 0, 250, np.array([0x3f800000], dtype=np.uint32))
                w.handle_requests([bg_SaveGameState(fan.confirmed)])
            try:
                if pipelined: fan.step_pipelined()          # results arrive later, in order: fan.results
                else: out.append(fan.step())
            except DesyncDetected as e:
                err = ("desync", e.frame, len(set(e.checksums)))
                break
        if err is None and pipelined:
            try: fan.drain()
            except DesyncDetected as e: err = ("desync", e.frame, len(set(e.checksums)))
            out = list(fan.results)
        if err is None:
            fan.settle()                           # back to the confirmed frame: identical on every rank
        q.put((rank, out, err, cm.snapshot_state(w, ids) if err is None else None))
    finally:
        dist.destroy_process_group()


def _run(world_size, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    args = (world_size, port, kw["n"], kw["D"], kw["bpr"], kw["steps"], kw.get("corrupt", 0), q, kw.get("pipelined", 0))
    procs = [ctx.Process(target=_worker, args=(r,) + args) for r in range(world_size)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _serial_reference(n, D, n_branches, steps, branch_input=None, confirmed_input=None, ttl_init=25, rate=50, warm=3):
    """The same fan-out computed by ONE process walking every branch (no collectives)."""
    import common as cm
    branch_input = branch_input or (lambda b, f: cm.INPUT_SPAWN if b % 2 == 0 else 0)
    confirmed_input = confirmed_input or (lambda f: cm.INPUT_SPAWN if f % 2 == 1 else 0)
    import common as cm
    from oracle.binding import OracleWorld
    import bevy_ggrs_amd as bg
    cap = n + 2 * rate * (steps + D + 2) * 2
    w = OracleWorld(cap, D + 1)
    ids = cm.build_particles(w, with_spawn=True, ttl_init=ttl_init)
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    cm.spawn_particles(w, ids, n, vel, ttl)
    for _ in range(warm):
        w.advance((0,))
    w.set_depth(D + 1)
    fn = cm.frame_spawn_fn(rate)

    def adv(frame, inp):
        a = bg.AdvanceFrame((inp,))
        if inp & cm.INPUT_SPAWN:
            a.spawn_vx, a.spawn_vy = fn(frame)
        return a
    w.set_confirmed(w.frame)
    w.handle_requests([bg.SaveGameState(w.frame)])
    C = w.frame
    out = []
    for _ in range(steps):
        per_branch = {}
        w.set_confirmed(C)
        for b in range(n_branches):
            # the confirmed input of frame C, then the branch's predicted inputs (fanout.py step())
            reqs = [bg.LoadGameState(C), adv(C, confirmed_input(C)), bg.SaveGameState(C + 1)]
            for i in range(1, D):
                reqs += [adv(C + i, branch_input(b, C + i)), bg.SaveGameState(C + 1 + i)]
            reqs.append(adv(C + D, branch_input(b, C + D)))
            per_branch[b] = w.handle_requests(reqs)
        assert len({v[0] for v in per_branch.values()}) == 1
        C += 1
        out.append({"confirmed_frame": C, "confirmed_checksum": per_branch[0][0], "branch_checksums": per_branch})
    w.handle_requests([bg.LoadGameState(C)])
    return out, cm.snapshot_state(w, ids)


@pytest.mark.parametrize("bpr,pipelined", [(1, 0), (3, 0), (1, 3), (3, 3), (5, 1)])
def test_fanout_two_ranks_matches_serial_walk(bpr, pipelined):
    """pipelined = desync_detection_interval of the pipelined path (0: one synchronous step at a time): the request list is the patched
    template (spawn payloads included), two steps are in flight and `pipelined` steps share one all-gather."""
    n, D, steps = 700, 4, 6 if not pipelined else 7      # (7 steps: the last all-gather group is a partly filled one)
    res = _run(2, n=n, D=D, bpr=bpr, steps=steps, pipelined=pipelined)
    ref, ref_state = _serial_reference(n, D, 2 * bpr, steps)
    import common as cm
    for rank, out, err, state in res:
        assert err is None
        assert len(out) == steps
        for got, want in zip(out, ref):
            assert got["confirmed_frame"] == want["confirmed_frame"]
            assert got["confirmed_checksum"] == want["confirmed_checksum"]
            assert got["branch_checksums"] == want["branch_checksums"]      # every rank sees every branch
        cm.assert_states_equal(state, ref_state, f"rank {rank}")
    # every branch agrees on the confirmed frame; branches with different predicted inputs diverge after it,
    # branches with the same inputs agree
    last = res[0][1][-1]["branch_checksums"]
    assert last[0][0] == last[1][0] and last[0][1:] != last[1][1:]
    if bpr >= 3:
        assert last[0] == last[2] == last[4] and last[1] == last[3] == last[5]


def _adopt_worker(rank, world_size, port, n, D, bpr, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        import common as cm
        import fanout_scenarios as fs
        from bevy_ggrs_amd.fanout import SpeculativeFanout
        from oracle.binding import OracleWorld
        cap = n + 100 * (steps * D + D + 8) * 2
        w = OracleWorld(cap, D + 1)
        ids = fs.build_world(w, n, rank == 0)
        fan = SpeculativeFanout(w, dist, D, HostExchange(w, ids), branches_per_rank=bpr, branch_input=fs.branch_input, confirmed_input=fs.true_input,
                                spawn_fn=cm.frame_spawn_fn(fs.RATE))
        seen = fs.run_adopt_session(fan, world_size * bpr, steps)
        for _ in range(2):
            o = fan.step(); seen.append((o["confirmed_frame"], o["confirmed_checksum"]))
        fan.settle()
        q.put((rank, seen, cm.snapshot_state(w, ids)))
    finally:
        dist.destroy_process_group()


def test_adopt_two_ranks_matches_the_straight_line_simulation():
    """SpeculativeFanout.adopt's control flow at world size 2 over gloo (oracle worlds keep no branch states: every rank re-simulates the adopted frames,
    what the non-owning ranks of the C-ABI path do): confirm -- speculate -- adopt rounds under a scripted true-input sequence end, on every rank, in the
    state and with the checksums of ONE world that simulated the true inputs frame by frame.  tests/test_gpu_zfanout.py runs the same script through
    ggrs_hip_fanout_adopt at world sizes 1, 2, 3 and 8."""
    import common as cm
    import fanout_scenarios as fs
    n, D, bpr, steps = 500, 5, 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=_adopt_worker, args=(r, 2, port, n, D, bpr, steps, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    seen0 = res[0][1]
    assert any(b - a > 1 for (a, _), (b, _) in zip(seen0, seen0[1:])), "no adoption jumped more than one frame"
    cap = n + 100 * (steps * D + D + 8) * 2
    cs, _, _ = fs.straight_line_reference(n, max(f for f, _ in seen0) - 3 + 1, cap)
    for rank, seen, state in res:
        assert seen == seen0
        for f, c in seen:
            assert c is None or cs[f] == c, (rank, f)
        cm.assert_states_equal(state, fs.state_at(n, state["frame"], cap), f"rank {rank}")


def test_fanout_detects_replica_desync():
    res = _run(2, n=300, D=3, bpr=1, steps=5, corrupt=2)
    for rank, out, err, _ in res:
        assert err is not None and err[0] == "desync" and err[2] == 2, (rank, err)
        assert len(out) == 2
