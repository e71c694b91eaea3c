"""Speculative re-simulation fan-out across the GPUs of one node (BASELINE.json config 5).

The reference has no multi-process story: one timeline's D resimulated
frames are a sequential chain.  What shards naturally is *speculation*: B predicted-input
branches, all starting from the same confirmed-frame snapshot, are independent.  One process
per GPU (torch.distributed, backend "nccl" == RCCL over xGMI):

  sync_confirmed()   rank `src` broadcasts its packed live state block (header + liveness and
                     presence masks + every registered word column; one contiguous buffer,
                     include/ggrs_hip.h `ggrs_hip_live_state_ptr`) -- ONE RCCL broadcast.  xGMI is
                     point-to-point, so this costs ~state_bytes / per-link bandwidth; it is done at
                     start-up and after a detected desync, NOT every step.
  step()             the true input of the confirmed frame C has arrived.  Every rank runs ONE request list:
                       [Load(C), Advance(confirmed input), Save(C+1)]                      once -- the prefix every branch shares
                       [Load(C+1), (Advance(predicted input of branch b), Save(C+1+i)) x (D-1), Advance(predicted)]   per branch b
                     Frame C+1 is the new confirmed frame -- rollback netcode's determinism keeps it
                     bit-identical on every rank, so moving the 1-byte input replaces re-broadcasting
                     60 B/entity -- and frames C+2.. are the branch's speculative future.  A branch's D
                     Checksum(u128)s are the shared C+1 entry followed by its own D-1; ONE all-gather carries
                     them, and the C+1 entries double as cross-rank desync detection (GgrsEvent::DesyncDetected
                     analogue).  (share_prefix=False: every branch replays Load(C), Advance, Save(C+1) itself --
                     1/D of the hashing done B times over, the round-3 shape.)
  step_pipelined()   the same, with one step in flight: step k+1 is enqueued on the device
                     (ggrs_hip_enqueue_requests) before step k's checksums are collected and
                     all-gathered on a side stream, so the collective and the host work overlap
                     the next step's kernel.

  adopt(branch, k)   the true inputs of the next k frames have arrived and equal what `branch` predicted: that branch's
                     retained state of frame C+k becomes the confirmed world (retain="all" / "newest") -- a ring-slot swap on
                     the rank that ran the branch; every other rank re-simulates the k frames with the now-confirmed inputs
                     (one launch, no bytes over xGMI, and a free desync check) or receives the block by ONE broadcast
                     (ggrs_hip_fanout_adopt; SURVEY.md 8e "the matching branch's state is adopted").

No data-path collective touches entity columns inside step(); per-GPU work is fixed as the
world size grows (weak scaling).  The collectives live INSIDE libggrs_hip.so (`native`: RcclFanout); `exchange` is the
host-side stand-in with the same contract that lets the same control flow run on gloo with CPU test worlds in
tests/test_fanout_gloo.py.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from .requests import AdvanceFrame, LoadGameState, SaveGameState


class DesyncDetected(RuntimeError):
    """Replicas of the confirmed frame disagree (GgrsEvent::DesyncDetected analogue,
    examples/stress_tests/particles.rs:299-314)."""

    def __init__(self, frame: int, checksums: Sequence[int]):
        super().__init__(f"confirmed frame {frame}: replica checksums differ: {[hex(c) for c in checksums]}")
        self.frame = frame
        self.checksums = list(checksums)


class RcclFanout:
    """The C ABI's fan-out entry points (include/ggrs_hip.h `ggrs_hip_fanout_*`): RCCL is called INSIDE libggrs_hip.so
    (ncclBroadcast of the packed live block, ncclAllGather of a step's checksums on a side stream); the host only
    carries the 128-byte ncclUniqueId from rank 0 to the other ranks."""

    def __init__(self, world, rank: int, world_size: int, unique_id: bytes):
        import ctypes as C
        from . import _ffi
        assert len(unique_id) == _ffi.FANOUT_ID_BYTES
        self._lib, self.world, self.rank, self.size = _ffi.lib, world, rank, world_size
        idb = (C.c_uint8 * _ffi.FANOUT_ID_BYTES).from_buffer_copy(unique_id)
        p = C.c_void_p()
        rc = self._lib.ggrs_hip_fanout_init(world._p, idb, rank, world_size, C.byref(p))
        if rc != 0:
            raise _ffi.GgrsHipError(rc, (self._lib.ggrs_hip_last_error(world._p) or b"").decode())
        self._p = p

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _ffi
        buf = (C.c_uint8 * _ffi.FANOUT_ID_BYTES)()
        rc = _ffi.lib.ggrs_hip_fanout_unique_id(buf)
        if rc != 0:
            raise _ffi.GgrsHipError(rc, "ncclGetUniqueId failed (is librccl.so loadable?)")
        return bytes(buf)

    def _check(self, rc: int):
        if rc != 0:
            from . import _ffi
            raise _ffi.GgrsHipError(rc, (self._lib.ggrs_hip_fanout_last_error(self._p) or b"").decode())

    def sync_confirmed(self, root: int = 0):
        self._check(self._lib.ggrs_hip_fanout_sync_confirmed(self._p, root))

    def step(self, requests) -> int:
        import ctypes as C
        arr, keep, n_save = self.world.build_requests(requests)
        got = C.c_uint32(0)
        self._check(self._lib.ggrs_hip_fanout_step(self._p, arr, len(requests), C.byref(got)))
        return got.value

    def step_raw(self, arr, n: int):
        self._check(self._lib.ggrs_hip_fanout_step(self._p, arr, n, None))

    def step_branches(self, bs):
        """ggrs_hip_fanout_step_branches: a prefix request list + n_branches x n_frames predicted inputs (a `_ffi.BranchStep`)."""
        import ctypes as C
        self._check(self._lib.ggrs_hip_fanout_step_branches(self._p, C.byref(bs), None))

    def adopt(self, branch: int, frame: int, replay=None, mode: int = 0) -> list:
        """ggrs_hip_fanout_adopt (collective): `branch`'s retained state of `frame` becomes the world.  replay: the request objects a rank that
        does NOT own the branch runs instead (mode 0); returns the Checksum(u128)s of its SaveGameStates ([] on the owner)."""
        import ctypes as C
        arr, keep, n_save = self.world.build_requests(replay or [])
        out = (C.c_uint64 * max(2, 2 * n_save))()
        got = C.c_uint32(0)
        self._check(self._lib.ggrs_hip_fanout_adopt(self._p, branch, frame, mode, arr if replay else None, len(replay or []), out, C.byref(got)))
        return [int(out[2 * i]) | (int(out[2 * i + 1]) << 64) for i in range(got.value)]

    def comm_info(self):
        """(rank, world size, HIP device) as the communicator itself reports them (ncclCommUserRank / ncclCommCount)."""
        import ctypes as C
        r, n, d = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.ggrs_hip_fanout_comm_info(self._p, C.byref(r), C.byref(n), C.byref(d)))
        return r.value, n.value, d.value

    def set_interval(self, steps_per_all_gather: int):
        self._check(self._lib.ggrs_hip_fanout_set_interval(self._p, steps_per_all_gather))

    def collect(self, max_u128_per_rank: int = 4096) -> np.ndarray:
        """Oldest all-gather group in flight -> (world_size, n_steps, n_saves, 2) u64 array of {lo, hi} per Save."""
        import ctypes as C
        if getattr(self, "_out", None) is None or self._out.size < self.size * max_u128_per_rank * 2:
            self._out = np.zeros(self.size * max_u128_per_rank * 2, dtype=np.uint64)
        steps, saves = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.ggrs_hip_fanout_collect(self._p, self._out.ctypes.data_as(C.POINTER(C.c_uint64)), max_u128_per_rank, C.byref(steps), C.byref(saves)))
        return self._out[: self.size * steps.value * saves.value * 2].reshape(self.size, steps.value, saves.value, 2).copy()

    def close(self):
        if getattr(self, "_p", None):
            self._lib.ggrs_hip_fanout_destroy(self._p)
            self._p = None

    def __del__(self):
        try: self.close()
        except Exception: pass


def C_cast_f32(addr: int):
    import ctypes as C
    return C.cast(C.c_void_p(addr), C.POINTER(C.c_float))


def default_branch_input(branch: int, frame: int) -> int:
    """256 distinct predicted-input sequences: the branch id repeated every frame."""
    return branch & 0xFF


default_branch_input.frame_invariant = True       # same byte for every frame: the driver asks once per branch, not once per branch and step


class SpeculativeFanout:
    def __init__(self, world, dist, depth: int, exchange, branches_per_rank: int = 1,
                 branch_input: Callable[[int, int], int] = default_branch_input,
                 confirmed_input: Callable[[int], int] = lambda frame: 0,
                 spawn_fn: Optional[Callable[[int], tuple]] = None, spawn_mask: int = 1 << 4,
                 num_players: int = 1, max_inflight: int = 1, desync_detection_interval: int = 1, native: "Optional[RcclFanout]" = None,
                 share_prefix: bool = True, retain: str = "none", compact: bool = True):
        self.w, self.dist, self.D, self.x = world, dist, depth, exchange
        self.native = native                                 # collectives inside libggrs_hip.so instead of torch.distributed (`exchange` unused)
        self.interval = max(1, desync_detection_interval)    # steps whose checksums share one all-gather (pipelined path)
        if native is not None:
            native.set_interval(self.interval)
        self._acc: list = []
        self.max_inflight = max_inflight                     # steps enqueued on the device before the oldest is collected
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.bpr = branches_per_rank
        # with ONE branch per rank there is nothing to share: the prefix as its own group would cost a second launch and a save / re-load of
        # C+1 (measured: 62.7 -> 87 us per step at 1 M, profiles/r04d/bench_fanout_ws1.json), so the list stays one fused group
        self.retain = {"none": 0, "newest": 2, "all": 4}[retain]    # GGRS_BRANCH_RETAIN_*: keep the branches' frames for adopt()
        self.share_prefix = bool(share_prefix) and depth >= 1 and (branches_per_rank > 1 or self.retain != 0)
        # the native path hands the library ONE compact description of the step (ggrs_hip_fanout_step_branches) instead of a request list of
        # ~4 x branches x depth entries; compact=False keeps the list (what rounds 3-5 measured; retention needs the compact form)
        self.compact = bool(compact) and native is not None and self.share_prefix
        assert not self.retain or self.compact or native is None, "retained branch states need the compact native step"
        self._bt = None
        # SaveGameState requests of one step's list on this rank == Checksum(u128)s it contributes to the all-gather
        self.saves_per_step = 1 + branches_per_rank * (depth - 1) if self.share_prefix else branches_per_rank * depth
        self.branch_input, self.confirmed_input = branch_input, confirmed_input
        self.spawn_fn, self.spawn_mask = spawn_fn, spawn_mask
        self._spawn_cache: dict = {}
        self.num_players = num_players
        self.synced = False
        self._last_raw = None
        self._tmpl = None
        self.confirmed = world.frame
        self._inflight: List[int] = []
        self._gathers: list = []
        self.results: list = []                              # every result the pipelined path completes, in step order
        # the gathered table of the next `raw_keep` completed steps, undecoded: (C, (size, bpr * D, 2) u64 array).  bench.py's parity
        # gate reads it AFTER its timed region (no Python ints per checksum while the clock runs)
        self.raw_keep, self.raw = 0, []
        world.set_depth(depth + 1)

    # ------------------------------------------------------------------ helpers
    def branch_ids(self, rank: Optional[int] = None) -> List[int]:
        r = self.rank if rank is None else rank
        return [r * self.bpr + j for j in range(self.bpr)]      # contiguous block per rank

    def _advance(self, frame: int, inp: int) -> AdvanceFrame:
        a = AdvanceFrame((inp,) * self.num_players)
        if self.spawn_fn is not None and (inp & self.spawn_mask):
            # pure function of the frame (the rolled-back RNG): every branch that spawns in frame f draws the same payload, so it is
            # drawn once per frame, not once per branch (a 256-branch step asks 1024 times for 8 distinct frames)
            pay = self._spawn_cache.get(frame)
            if pay is None:
                if len(self._spawn_cache) > 64: self._spawn_cache.clear()
                pay = self._spawn_cache[frame] = self.spawn_fn(frame)
            a.spawn_vx, a.spawn_vy = pay
        return a

    def sync_confirmed(self, src: int = 0):
        """Broadcast `src`'s live world as the confirmed frame C and snapshot it on every rank."""
        self.drain()
        if self.native is not None:
            self.native.sync_confirmed(src)
        else:
            self.x.broadcast(self.dist, src)
        self.w.set_confirmed(self.w.frame)
        cs = self.w.handle_requests([SaveGameState(self.w.frame)])[0]
        self.confirmed = self.w.frame
        self.synced = True
        return cs

    # ------------------------------------------------------------------ one confirmed frame
    def _requests(self, C: int) -> list:
        D = self.D
        c_in = self.confirmed_input(C)
        reqs: list = []
        if self.share_prefix:
            # the branch-invariant prefix, once: the confirmed input takes frame C to C+1 and C+1 is saved -- it is what every branch
            # starts from AND the snapshot the next step loads
            reqs += [LoadGameState(C), self._advance(C, c_in), SaveGameState(C + 1)]
            for b in self.branch_ids():
                reqs.append(LoadGameState(C + 1))
                for i in range(1, D):
                    reqs += [self._advance(C + i, self.branch_input(b, C + i)), SaveGameState(C + 1 + i)]
                reqs.append(self._advance(C + D, self.branch_input(b, C + D)))  # the newest predicted frame stays live-only
            return reqs
        for b in self.branch_ids():
            reqs += [LoadGameState(C), self._advance(C, c_in), SaveGameState(C + 1)]
            for i in range(1, D):
                reqs += [self._advance(C + i, self.branch_input(b, C + i)), SaveGameState(C + 1 + i)]
            reqs.append(self._advance(C + D, self.branch_input(b, C + D)))      # the newest predicted frame stays live-only
        return reqs

    def _expand(self, allv: np.ndarray) -> np.ndarray:
        """What the ranks gathered -> the canonical table (size, bpr * D, 2): every branch's D checksums, C+1 first."""
        D, bpr = self.D, self.bpr
        if not self.share_prefix:
            return allv.reshape(self.size, max(bpr * D, 1), 2)
        a = allv.reshape(self.size, self.saves_per_step, 2)
        conf = np.broadcast_to(a[:, 0:1][:, None], (self.size, bpr, 1, 2))
        own = a[:, 1:].reshape(self.size, bpr, D - 1, 2)
        return np.ascontiguousarray(np.concatenate([conf, own], axis=2)).reshape(self.size, bpr * D, 2)

    def _check(self, C: int, allv: np.ndarray, want_result: bool = True) -> Optional[dict]:
        D, n = self.D, self.bpr * self.D
        allv = self._expand(allv)
        conf = allv[:, 0:n:D, :].reshape(-1, 2)              # the C+1 entry of every branch of every rank
        if (conf != conf[0]).any():
            self.synced = False                              # caller may sync_confirmed() again
            raise DesyncDetected(C + 1, [int(p[0]) | (int(p[1]) << 64) for p in conf])
        self._last_raw = (C, allv)
        if len(self.raw) < self.raw_keep:
            self.raw.append((C, allv.copy()))
        return self.last if want_result else None

    def _finish(self, C: int, mine: np.ndarray, want_result: bool = True, defer: bool = False) -> Optional[dict]:
        """mine: this rank's bpr x D checksums as a (bpr*D, 2) u64 array {lo, hi}; u128 as 2 x u64.
        Synchronous path: ONE all-gather per step.  Pipelined path (defer): the checksums of
        `desync_detection_interval` consecutive steps travel in ONE all-gather (the reference's stress_test exchanges
        checksums every `--desync-detection-interval` frames, default 10: examples/stress_tests/particles.rs:49,135-137),
        and the host never waits for a collective it has just launched: it completes the one started before."""
        if not defer:
            return self._check(C, self.x.all_gather_u64(self.dist, np.ascontiguousarray(mine).reshape(-1)), want_result)
        self._acc.append((C, np.ascontiguousarray(mine)))
        if len(self._acc) < self.interval:
            return None
        return self._launch_gather(want_result)

    def _launch_gather(self, want_result: bool) -> Optional[dict]:
        steps = [c for c, _ in self._acc]
        flat = np.concatenate([m.reshape(-1) for _, m in self._acc])
        self._acc = []
        if hasattr(self.x, "all_gather_u64_start"):
            self._gathers.append((steps, self.x.all_gather_u64_start(self.dist, flat)))
            if len(self._gathers) <= 1:
                return None
            return self._complete(*self._gathers.pop(0), want_result)
        return self._complete(steps, None, want_result, data=self.x.all_gather_u64(self.dist, flat))

    def _complete(self, steps, token, want_result: bool, data=None) -> Optional[dict]:
        allv = data if data is not None else self.x.all_gather_u64_finish(token)
        allv = allv.reshape(self.size, len(steps), -1)
        out = None
        for k, C in enumerate(steps):
            out = self._check(C, np.ascontiguousarray(allv[:, k, :]), want_result)
            if out is not None:
                self.results.append(out)
        return out

    @property
    def last(self) -> dict:
        if self._last_raw is None:
            return {}
        C, allv = self._last_raw
        D = self.D
        to_int = lambda p: int(p[0]) | (int(p[1]) << 64)
        return {"confirmed_frame": C + 1, "confirmed_checksum": to_int(allv[0, 0]),
                "branch_checksums": {r * self.bpr + j: [to_int(allv[r, j * D + i]) for i in range(D)]
                                     for r in range(self.size) for j in range(self.bpr)}}

    @staticmethod
    def _as_u64_pairs(cs: Sequence[int]) -> np.ndarray:
        out = np.zeros((max(len(cs), 1), 2), dtype=np.uint64)
        for k, c in enumerate(cs):
            out[k] = (c & 0xFFFFFFFFFFFFFFFF, c >> 64)
        return out

    def step(self, want_result: bool = True) -> Optional[dict]:
        """One confirmed frame, synchronously."""
        if not self.synced:
            self.sync_confirmed(0)
        self.drain()
        if self.native is not None:
            self._native_enqueue()
            return self._native_collect(want_result)        # closes the (partly filled) group: one all-gather for this step
        C = self.confirmed
        self.w.set_confirmed(C)                              # discard_old_snapshots bound
        cs = self.w.handle_requests(self._requests(C))
        self.confirmed = C + 1
        return self._finish(C, self._as_u64_pairs(cs), want_result)

    # ---- pre-marshalled request list for the pipelined path: a step's list differs from the previous one only in its frame numbers,
    # input bytes and -- with a spawn system -- in which payload each spawning AdvanceFrame points at; all of it is patched in place
    # (marshalling ~4000 request objects per step was 6 of the 7.4 ms of a 256-branch step with diverging branches, profiles/r04i)
    def _template(self):
        import ctypes as C_
        spawn_fn, self.spawn_fn = self.spawn_fn, None        # the template carries no payload: _patch points the spawning advances at theirs
        try: reqs = self._requests(0)
        finally: self.spawn_fn = spawn_fn
        arr, keep, n_save = self.w.build_requests(reqs)
        loads = [(i, r.frame) for i, r in enumerate(reqs) if isinstance(r, LoadGameState)]
        saves = [(i, r.frame) for i, r in enumerate(reqs) if isinstance(r, SaveGameState)]
        advs = [i for i, r in enumerate(reqs) if isinstance(r, AdvanceFrame)]
        out = (C_.c_uint64 * (2 * max(n_save, 1)))()
        t = {"arr": arr, "keep": keep, "n": len(reqs), "n_save": n_save, "loads": loads, "saves": saves, "advs": advs,
             "out": out, "out_np": np.frombuffer(out, dtype=np.uint64).reshape(-1, 2)}
        # frame of every AdvanceFrame relative to C, in list order (a Load sets the frame, an Advance moves it on by one)
        rel, cur = [], 0
        for r in reqs:
            if isinstance(r, LoadGameState): cur = r.frame
            elif isinstance(r, AdvanceFrame): rel.append(cur); cur += 1
        t["adv_rel"] = np.array(rel, dtype=np.intp)
        # the frame field of every Load / Save as one strided view: a step rewrites them with one assignment
        R = type(arr[0]); sz = C_.sizeof(R)
        t["frame_view"] = np.ndarray((len(reqs),), dtype=np.int32, buffer=arr, offset=R.frame.offset, strides=(sz,))
        t["frame_idx"] = np.array([i for i, _ in loads + saves], dtype=np.intp)
        t["frame_rel"] = np.array([f for _, f in loads + saves], dtype=np.int64)
        if spawn_fn is not None:
            # the three spawn fields of every request as strided views of the ctypes array (pointers as u64): one assignment per field per step
            view = lambda field: np.ndarray((len(reqs),), dtype=np.uint64, buffer=arr, offset=getattr(R, field).offset, strides=(sz,))
            t["spawn_views"] = (view("spawn_count"), view("spawn_vx"), view("spawn_vy"))
            t["adv_idx"] = np.array(advs, dtype=np.intp)
        return t

    def _payload(self, frame: int):
        """(count, address of vx, address of vy, arrays) of the frame's spawn payload: a pure function of the frame, drawn once."""
        pay = self._spawn_cache.get(frame)
        if pay is None:
            if len(self._spawn_cache) > 64: self._spawn_cache.clear()
            pay = self._spawn_cache[frame] = self.spawn_fn(frame)
        vx, vy = (np.ascontiguousarray(a, dtype=np.float32) for a in pay)
        if vx is not pay[0] or vy is not pay[1]: self._spawn_cache[frame] = (vx, vy)
        return vx.size, vx.ctypes.data, vy.ctypes.data, (vx, vy)

    def _patch(self, t, C: int):
        arr, D = t["arr"], self.D
        t["frame_view"][t["frame_idx"]] = (t["frame_rel"] + C).astype(np.int32)
        c_in = self.confirmed_input(C)
        ids = self.branch_ids()
        pred = t.get("pred") if getattr(self.branch_input, "frame_invariant", False) else None
        if self.share_prefix:                                # advance 0: the confirmed input; then D predicted ones per branch (frames C+1 .. C+D)
            if pred is None: pred = [self.branch_input(ids[(k - 1) // D], C + 1 + (k - 1) % D) for k in range(1, len(t["advs"]))]
            inputs = [c_in] + pred
        else:
            per_branch = D + 1
            if pred is None: pred = [0 if k % per_branch == 0 else self.branch_input(ids[k // per_branch], C + k % per_branch) for k in range(len(t["advs"]))]
            inputs = list(pred); inputs[::per_branch] = [c_in] * len(inputs[::per_branch])
        t["pred"] = pred
        if inputs != t.get("inputs"):                        # (the usual case: predictions repeat, nothing to rewrite)
            t["inputs"] = inputs
            t["spawning"] = (np.array(inputs, dtype=np.int64) & self.spawn_mask) != 0
            for k, i in enumerate(t["advs"]):
                q = arr[i]
                for p in range(q.n_inputs):
                    q.inputs[p] = inputs[k]
        if self.spawn_fn is not None:
            # every spawning AdvanceFrame points at the payload of ITS frame (C + its relative frame): D + 1 payloads per step at most,
            # the library copies them into its own ring while it enqueues (they only have to live until that call returns)
            rels = np.unique(t["adv_rel"][t["spawning"]])
            cnt = np.zeros(D + 2, dtype=np.uint64); px = np.zeros(D + 2, dtype=np.uint64); py = np.zeros(D + 2, dtype=np.uint64)
            t["pay_keep"] = []
            for r in rels:
                cnt[r], px[r], py[r], keep = self._payload(C + int(r)); t["pay_keep"].append(keep)
            vc, vx, vy = t["spawn_views"]
            sel, idx = t["adv_rel"], t["adv_idx"]
            vc[idx] = np.where(t["spawning"], cnt[sel], 0); vx[idx] = np.where(t["spawning"], px[sel], 0); vy[idx] = np.where(t["spawning"], py[sel], 0)

    def step_pipelined(self, want_result: bool = True) -> Optional[dict]:
        """Enqueue the step of confirmed frame C on the device, THEN collect and all-gather the oldest step once more
        than `max_inflight` are queued.  Returns that step's result (None while the pipeline fills)."""
        if not self.synced:
            self.sync_confirmed(0)
        if self.native is not None:
            self._native_enqueue()
            # keep one whole all-gather group (+ max_inflight steps) in flight: the host never waits for a collective it has just issued
            return self._native_collect(want_result) if len(self._inflight) >= self.interval + self.max_inflight else None
        if self.saves_per_step > 256 or not hasattr(self.w, "enqueue_requests_raw"):
            return self.step(want_result)
        C = self.confirmed
        self.w.set_confirmed(C)
        if self._tmpl is None:
            self._tmpl = self._template()
        self._patch(self._tmpl, C)
        self.w.enqueue_requests_raw(self._tmpl["arr"], self._tmpl["n"])
        self._inflight.append(C)
        self.confirmed = C + 1
        return self._collect_one(want_result) if len(self._inflight) > self.max_inflight else None

    # ---- native path: ggrs_hip_fanout_step (enqueue + all-gather on a side stream inside the library) / _collect
    # ---- the compact native step: prefix [Load(C), Advance(confirmed), Save(C+1)] + ONE table of predicted inputs [branch][frame][player bytes]
    def _branch_template(self):
        import ctypes as C_
        from . import _ffi
        D, bpr, ib, npl = self.D, self.bpr, getattr(self.w, "input_bytes", 1), self.num_players
        spawn_fn, self.spawn_fn = self.spawn_fn, None
        try: pre = [LoadGameState(0), self._advance(0, 0), SaveGameState(1)]
        finally: self.spawn_fn = spawn_fn
        arr, keep, _ = self.w.build_requests(pre)
        t = {"arr": arr, "keep": keep, "bs": _ffi.BranchStep(), "inputs": np.zeros((bpr, D, npl * ib), dtype=np.uint8), "sel": np.zeros((bpr, D), dtype=np.uint16),
             "table": (_ffi.BranchSpawn * D)(), "pred": None}
        bs = t["bs"]
        bs.prefix, bs.n_prefix, bs.n_branches, bs.n_frames, bs.n_inputs, bs.flags = arr, 3, bpr, D, npl, self.retain
        bs.inputs = t["inputs"].ctypes.data
        if spawn_fn is not None:
            bs.spawn_table, bs.n_spawn_table, bs.spawn_sel = t["table"], D, t["sel"].ctypes.data
        return t

    def _native_enqueue_compact(self):
        C = self.confirmed
        self.w.set_confirmed(C)
        if self._bt is None:
            self._bt = self._branch_template()
        t, D, ib = self._bt, self.D, getattr(self.w, "input_bytes", 1)
        arr = t["arr"]
        arr[0].frame, arr[2].frame = C, C + 1
        c_in = self.confirmed_input(C)
        q = arr[1]
        for p in range(q.n_inputs): q.inputs[p * ib] = c_in
        keep = []
        if self.spawn_fn is not None:
            if c_in & self.spawn_mask:
                cnt, px, py, k = self._payload(C); keep.append(k)
                q.spawn_count = cnt
                q.spawn_vx = C_cast_f32(px); q.spawn_vy = C_cast_f32(py)
            else:
                q.spawn_count = 0; q.spawn_vx = None; q.spawn_vy = None
        ids = self.branch_ids()
        if t["pred"] is None or not getattr(self.branch_input, "frame_invariant", False):
            # advance i of a branch takes frame C+1+i to C+2+i with the branch's predicted input of frame C+1+i
            pred = np.array([[self.branch_input(b, C + 1 + i) for i in range(D)] for b in ids], dtype=np.uint8)
            if t["pred"] is None or not np.array_equal(pred, t["pred"]):
                t["pred"] = pred
                t["inputs"][:, :, ::ib] = pred[:, :, None]
                t["sel"][:] = np.where((pred & self.spawn_mask) != 0, np.arange(1, D + 1, dtype=np.uint16)[None, :], 0) if self.spawn_fn is not None else 0
        if self.spawn_fn is not None:
            for i in np.unique(np.nonzero(t["sel"])[1]) if t["sel"].any() else ():
                cnt, px, py, k = self._payload(C + 1 + int(i)); keep.append(k)
                e = t["table"][int(i)]
                e.count, e.vx, e.vy = cnt, px, py
        t["pay_keep"] = keep
        self.native.step_branches(t["bs"])
        self._inflight.append(C)
        self.confirmed = C + 1

    def _native_enqueue(self):
        if self.compact:
            return self._native_enqueue_compact()
        C = self.confirmed
        self.w.set_confirmed(C)
        if self._tmpl is None:
            self._tmpl = self._template()
        self._patch(self._tmpl, C)
        self.native.step_raw(self._tmpl["arr"], self._tmpl["n"])
        self._inflight.append(C)
        self.confirmed = C + 1

    def _native_collect(self, want_result: bool = True) -> Optional[dict]:
        allv = self.native.collect()                          # (size, n_steps, n_saves, 2): the oldest all-gather group
        out = None
        for k in range(allv.shape[1]):
            C = self._inflight.pop(0)
            out = self._check(C, np.ascontiguousarray(allv[:, k]).reshape(-1), want_result)
            if out is not None:
                self.results.append(out)
        return out

    def _collect_one(self, want_result: bool = True) -> Optional[dict]:
        if self.native is not None:
            return self._native_collect(want_result)
        C = self._inflight.pop(0)
        n = self.saves_per_step
        if self._tmpl is not None:
            self.w.collect_checksums_raw(self._tmpl["out"], n)
            return self._finish(C, self._tmpl["out_np"][:max(n, 1)].copy(), want_result, defer=True)
        return self._finish(C, self._as_u64_pairs(self.w.collect_checksums(n)), want_result, defer=True)

    def drain(self, want_result: bool = True) -> Optional[dict]:
        """Collect every step still on the device and complete every collective still in flight; returns the newest
        result.  (Results of the pipelined path arrive in order, possibly more than one per call: see `results`.)"""
        out = None
        while self._inflight:
            r = self._collect_one(want_result)
            if r is not None: out = r
        if self._acc:                                        # a partly filled interval
            r = self._launch_gather(want_result)
            if r is not None: out = r
        while self._gathers:
            r = self._complete(*self._gathers.pop(0), want_result)
            if r is not None: out = r
        return out

    def settle(self) -> int:
        """Bring the live world back to the confirmed frame (drops the speculative tail; the compact native step leaves it there anyway)."""
        self.drain()
        self.w.handle_requests([LoadGameState(self.confirmed)])
        return self.confirmed

    # ------------------------------------------------------------------ adoption
    def adopt(self, branch: int, k: int, broadcast: bool = False) -> int:
        """The true inputs of frames C .. C+k-1 (C = the confirmed frame) have arrived and equal what GLOBAL branch `branch` predicted: its state of
        frame C+k becomes the confirmed world on every rank.  Native + retain: ggrs_hip_fanout_adopt -- the owning rank swaps a ring slot for the
        branch's retained block, the others re-simulate the k frames (or, broadcast=True, receive the block).  Worlds without retained branches (the
        CPU test worlds of the gloo test): every rank re-simulates.  The re-simulated Checksum(u128) of frame C+k is compared with the one the branch
        delivered through the all-gather (DesyncDetected).  Collective; returns the new confirmed frame."""
        self.drain()
        C = self.confirmed
        assert 1 <= k <= self.D, (k, self.D)
        replay = [self._advance(C + i, self.branch_input(branch, C + i)) for i in range(k)] + [SaveGameState(C + k)]
        if self.native is not None and self.retain:
            cs = self.native.adopt(branch, C + k, replay, mode=1 if broadcast else 0)
        else:
            self.w.set_confirmed(C)
            cs = self.w.handle_requests([LoadGameState(C)] + replay)
            self.w.set_confirmed(C + k)
        want = None
        if self._last_raw is not None and self._last_raw[0] == C - 1 and k <= self.D - 1:
            # the last step started at confirmed frame C-1: branch entry 0 is frame C, entry k is frame C+k
            r, j = divmod(branch, self.bpr)
            p = self._last_raw[1][r, j * self.D + k]
            want = int(p[0]) | (int(p[1]) << 64)
        if cs and want is not None and cs[-1] != want:
            self.synced = False
            raise DesyncDetected(C + k, [cs[-1], want])
        self.confirmed = C + k
        self.adopted_checksum = want if want is not None else (cs[-1] if cs else None)
        return self.confirmed
