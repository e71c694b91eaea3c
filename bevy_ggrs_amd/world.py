"""Thin object wrapper over the C ABI (include/ggrs_hip.h): one `World` == one `ggrs_world`.

This is plumbing for the host mirror in `bevy_ggrs_amd.app`; every data-path operation is a
HIP kernel launch inside libggrs_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Mapping, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import GgrsHipError
from .requests import AdvanceFrame, LoadGameState, SaveGameState


def _as_c(arr: np.ndarray):
    return arr.ctypes.data_as(C.c_void_p)


class WorldBase:
    """Backend-neutral surface.  `bevy_ggrs_amd.World` binds it to libggrs_hip.so; the CPU
    oracle (tests only) binds the same surface to oracle/_build/libggrs_oracle.so."""

    # ---- to be provided by the backend -------------------------------------------------
    _p: C.c_void_p
    _lib = None
    _prefix = ""
    _request_cls = _ffi.Request
    _sysdesc_cls = _ffi.SystemDesc

    def _fn(self, name):
        return getattr(self._lib, self._prefix + name)

    def _check(self, rc: int):
        if rc != 0:
            msg = self._fn("last_error")(self._p)
            raise GgrsHipError(rc, msg.decode() if msg else "")

    # ---- registration -------------------------------------------------------------------
    def register_component(self, name: str, word_bytes: int, n_words: int, rollback: bool = True) -> int:
        """rollback_component_with_copy (rollback=True), or a plain device-resident component that is
        NOT registered for rollback (rollback=False; see ggrs_hip_register_component_ex)."""
        cid = C.c_uint32(0)
        if rollback:
            self._check(self._fn("register_component")(self._p, name.encode(), word_bytes, n_words, C.byref(cid)))
        else:
            self._check(self._fn("register_component_ex")(self._p, name.encode(), word_bytes, n_words,
                                                          _ffi.COMP_NO_ROLLBACK, C.byref(cid)))
        self._comps.append((name, word_bytes, n_words))
        return cid.value

    def set_component_default(self, comp: int, words: np.ndarray):
        _, wb, nw = self._comps[comp]
        arr = np.ascontiguousarray(words).view(np.uint8).reshape(-1)
        assert arr.size == wb * nw, "default must be n_words words"
        self._check(self._fn("set_component_default")(self._p, comp, _as_c(arr)))

    def checksum_component(self, comp: int, word_idx: Sequence[int]):
        a = (C.c_uint32 * len(word_idx))(*word_idx)
        self._check(self._fn("checksum_component")(self._p, comp, a, len(word_idx)))

    def add_system(self, kind: int, comp=(), word=(), iparam=(), fparam=()):
        d = self._sysdesc_cls()
        d.kind = kind
        for i, v in enumerate(comp): d.comp[i] = v
        for i, v in enumerate(word): d.word[i] = v
        for i, v in enumerate(iparam): d.iparam[i] = v
        for i, v in enumerate(fparam): d.fparam[i] = v
        self._check(self._fn("add_system")(self._p, C.byref(d)))

    def set_frame_rate(self, fps: int):
        r = self._fn("set_frame_rate")(self._p, fps)
        if r is not None: self._check(r)

    # ---- entities -----------------------------------------------------------------------
    def spawn(self, count: int, comps: Mapping[int, Optional[Sequence[Optional[np.ndarray]]]]) -> int:
        """commands.spawn((.., Rollback)) x count.  comps: {comp_id: [word arrays] or None}."""
        mask = 0
        ptrs, keep = [], []
        for cid in sorted(comps):
            mask |= 1 << cid
            _, wb, nw = self._comps[cid]
            cols = comps[cid]
            for k in range(nw):
                a = None if cols is None else cols[k]
                if a is None:
                    ptrs.append(None)
                else:
                    a = np.ascontiguousarray(a)
                    assert a.itemsize == wb and a.size == count, (a.dtype, a.size, count)
                    keep.append(a)
                    ptrs.append(a.ctypes.data)
        arr = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        first = C.c_uint64(0)
        self._check(self._fn("spawn")(self._p, count, mask, arr, C.byref(first)))
        return first.value

    def despawn(self, slot: int):
        self._check(self._fn("despawn")(self._p, slot))

    def despawn_rollback(self, slot: int):
        """commands.entity(e).despawn_rollback() (snapshot/despawn.rs:114-143)."""
        self._check(self._fn("despawn_rollback")(self._p, slot))

    def disabled_mask(self, n_slots: Optional[int] = None) -> np.ndarray:
        """RollbackDespawned markers of the live world (peer-local, outside every snapshot)."""
        return self._mask("download_disabled", self.len if n_slots is None else n_slots)

    def despawned_frames(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        if count is None: count = self.len - first
        out = np.zeros(max(count, 1), dtype=np.int32)
        if count: self._check(self._fn("download_despawned_frames")(self._p, first, count, _as_c(out)))
        return out[:count]

    def insert_component(self, comp: int, slot: int, words: np.ndarray):
        arr = np.ascontiguousarray(words).view(np.uint8).reshape(-1)
        self._check(self._fn("insert_component")(self._p, comp, slot, _as_c(arr)))

    def remove_component(self, comp: int, slot: int):
        self._check(self._fn("remove_component")(self._p, comp, slot))

    def upload_word(self, comp: int, word: int, first: int, data: np.ndarray):
        _, wb, _ = self._comps[comp]
        a = np.ascontiguousarray(data)
        assert a.itemsize == wb
        self._check(self._fn("upload_word")(self._p, comp, word, first, a.size, _as_c(a)))

    def download_word(self, comp: int, word: int, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        _, wb, _ = self._comps[comp]
        if count is None: count = self.len - first
        out = np.empty(count, dtype={1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[wb])
        if count: self._check(self._fn("download_word")(self._p, comp, word, first, count, _as_c(out)))
        return out

    def _mask(self, fn_name, n_slots, *pre):
        nw = (n_slots + 63) // 64
        words = np.zeros(max(nw, 1), dtype=np.uint64)
        self._check(self._fn(fn_name)(self._p, *pre, _as_c(words), nw))
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_slots]
        return bits.astype(bool)

    def alive_mask(self, n_slots: Optional[int] = None) -> np.ndarray:
        return self._mask("download_alive", self.len if n_slots is None else n_slots)

    def present_mask(self, comp: int, n_slots: Optional[int] = None) -> np.ndarray:
        return self._mask("download_present", self.len if n_slots is None else n_slots, comp)

    @property
    def len(self) -> int:
        return int(self._fn("len")(self._p))

    def active_count(self) -> int:
        out = C.c_uint64(0)
        r = self._fn("active_count")(self._p, C.byref(out))
        self._check(r)
        return out.value

    # ---- counters / ring ----------------------------------------------------------------
    @property
    def frame(self) -> int:
        return int(self._fn("frame")(self._p))

    def set_frame(self, f: int):
        r = self._fn("set_frame")(self._p, f)
        if r is not None: self._check(r)

    def set_depth(self, d: int):
        r = self._fn("set_depth")(self._p, d)
        if r is not None: self._check(r)

    def set_confirmed(self, frame: Optional[int]):
        r = self._fn("set_confirmed")(self._p, 0 if frame is None else 1, 0 if frame is None else frame)
        if r is not None: self._check(r)

    def has_snapshot(self, frame: int) -> bool:
        return bool(self._fn("has_snapshot")(self._p, frame))

    def snapshot_count(self) -> int:
        return int(self._fn("snapshot_count")(self._p))

    # ---- requests -----------------------------------------------------------------------
    def save(self) -> int:
        out = (C.c_uint64 * 2)()
        self._check(self._fn("save")(self._p, out))
        return int(out[0]) | (int(out[1]) << 64)

    def load(self, frame: int):
        self._check(self._fn("load")(self._p, frame))

    def advance(self, inputs: Iterable[int] = (), dt_bits: int = 0, spawn_vx=None, spawn_vy=None):
        inp, n_players = self._input_bytes(tuple(inputs))
        ia = (C.c_uint8 * max(1, len(inp)))(*inp)
        n = 0
        vx = vy = None
        if spawn_vx is not None:
            vx = np.ascontiguousarray(spawn_vx, dtype=np.float32); vy = np.ascontiguousarray(spawn_vy, dtype=np.float32)
            n = vx.size
        self._check(self._fn("advance")(
            self._p, dt_bits, ia, n_players, n,
            vx.ctypes.data_as(C.POINTER(C.c_float)) if n else None,
            vy.ctypes.data_as(C.POINTER(C.c_float)) if n else None))

    input_bytes = 1

    def _input_bytes(self, inputs):
        """PlayerInputs -> (the bytes the ABI wants, number of players): one int per player for one-byte inputs, else `input_bytes` bytes per
        player as bytes objects / int sequences."""
        ib = self.input_bytes
        if ib == 1 and all(isinstance(x, (int, np.integer)) for x in inputs):
            b = bytes(int(x) & 0xFF for x in inputs)
            return b, len(b)
        out = b"".join(bytes(x) if not isinstance(x, (int, np.integer)) else int(x).to_bytes(ib, "little") for x in inputs)
        assert len(out) == ib * len(inputs), f"every player's input must be {ib} bytes"
        return out, len(inputs)

    def set_input_layout(self, input_bytes: int, max_players: int = 16):
        """PlayerInputs<T>: size_of::<T::Input>() and the players of the session (ggrs_hip_set_input_layout); before the first custom system."""
        self._check(self._fn("set_input_layout")(self._p, input_bytes, max_players))
        self.input_bytes = input_bytes

    def build_requests(self, requests):
        """Python request objects -> (ctypes array, keep-alive list, number of saves)."""
        n = len(requests)
        arr = (self._request_cls * max(1, n))()
        keep = []
        n_save = 0
        for i, r in enumerate(requests):
            q = arr[i]
            if isinstance(r, SaveGameState):
                q.kind, q.frame = _ffi.REQ_SAVE, r.frame
                n_save += 1
            elif isinstance(r, LoadGameState):
                q.kind, q.frame = _ffi.REQ_LOAD, r.frame
            elif isinstance(r, AdvanceFrame):
                q.kind = _ffi.REQ_ADVANCE
                q.dt_bits = r.dt_bits
                inp, n_players = self._input_bytes(r.inputs)
                if inp:
                    ia = (C.c_uint8 * len(inp))(*inp); keep.append(ia)
                    q.inputs = C.cast(ia, C.POINTER(C.c_uint8)); q.n_inputs = n_players
                if r.status is not None and n_players:
                    st = bytes(r.status); assert len(st) == n_players, "one InputStatus per player"
                    sa = (C.c_uint8 * len(st))(*st); keep.append(sa)
                    q.status = C.cast(sa, C.POINTER(C.c_uint8))
                if r.spawn_vx is not None and len(r.spawn_vx):
                    vx = np.ascontiguousarray(r.spawn_vx, dtype=np.float32); vy = np.ascontiguousarray(r.spawn_vy, dtype=np.float32)
                    keep += [vx, vy]
                    q.spawn_count = vx.size
                    q.spawn_vx = vx.ctypes.data_as(C.POINTER(C.c_float))
                    q.spawn_vy = vy.ctypes.data_as(C.POINTER(C.c_float))
                if r.spawn_count:
                    q.spawn_count = int(r.spawn_count)
                if r.spawn_payload is not None:
                    pl = np.ascontiguousarray(np.frombuffer(r.spawn_payload, dtype=np.uint8) if isinstance(r.spawn_payload, (bytes, bytearray)) else np.asarray(r.spawn_payload)).view(np.uint8).reshape(-1)
                    if pl.size:
                        keep.append(pl)
                        q.spawn_payload = pl.ctypes.data
                        q.spawn_payload_bytes = pl.size
            else:
                raise TypeError(r)
        return arr, keep, n_save

    def handle_requests(self, requests) -> list:
        """handle_requests (schedule_systems.rs:170-289): returns the Checksum(u128) of every
        SaveGameState, in request order."""
        arr, keep, n_save = self.build_requests(requests)
        out = (C.c_uint64 * max(2, 2 * n_save))()
        self._check(self._fn("handle_requests")(self._p, arr, len(requests), out))
        return [int(out[2 * i]) | (int(out[2 * i + 1]) << 64) for i in range(n_save)]

    def handle_requests_raw(self, arr, n: int, out):
        """Pre-built ctypes request array (bench hot loop: no Python per-request work)."""
        self._check(self._fn("handle_requests")(self._p, arr, n, out))


class World(WorldBase):
    """A device world on one MI355X.  Raises GgrsHipError(GGRS_E_NO_DEVICE) without a GPU."""
    _lib = _ffi.lib
    _prefix = "ggrs_hip_"

    def __init__(self, capacity: int, max_depth: int = 8, device: int = 0, stream: int = 0,
                 arena_ptr: int = 0, arena_bytes: int = 0, flags: int = 0):
        d = _ffi.WorldDesc()
        d.device, d.max_depth, d.capacity = device, max_depth, capacity
        d.stream = stream or None
        d.arena = arena_ptr or None
        d.arena_bytes = arena_bytes
        d.flags = flags
        p = C.c_void_p()
        rc = self._lib.ggrs_hip_world_create_ex(C.byref(d), C.byref(p))
        if rc != 0:
            raise GgrsHipError(rc, "world_create failed" + (": no HIP device visible (the product path has no CPU fallback)" if rc == _ffi.GGRS_E_NO_DEVICE else ""))
        self._p = p
        self._comps = []
        self.capacity = capacity

    def add_custom_system(self, source: str, bindings: Sequence[tuple], iparam=(), fparam=(), name: str = "custom"):
        """add_systems(GgrsSchedule, <your system>) for a per-entity system written in HIP C++ (ggrs_hip_add_custom_system):
        `source` defines `__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f)`, `bindings` = [(comp, word), ..]
        are the words it sees as e.f32(i)/e.u32(i)/e.i32(i)/e.u64(i).  Compiled for gfx950 when added; a compile error raises
        GgrsHipError carrying the compiler log."""
        d = _ffi.CustomSystemDesc()
        d.name, d.source, d.n_bindings = name.encode(), source.encode(), len(bindings)
        if len(bindings) > _ffi.CUSTOM_MAX_BINDINGS:
            raise ValueError(f"at most {_ffi.CUSTOM_MAX_BINDINGS} bindings")
        for i, (c, w) in enumerate(bindings): d.comp[i], d.word[i] = c, w
        for i, v in enumerate(iparam): d.iparam[i] = v
        for i, v in enumerate(fparam): d.fparam[i] = v
        self._check(self._lib.ggrs_hip_add_custom_system(self._p, C.byref(d)))

    def add_spawn_system(self, source: str, bundle: Sequence[int], bindings: Sequence[tuple] = (), payload_stride: int = 0, iparam=(), fparam=(), name: str = "spawn"):
        """add_systems(GgrsSchedule, <a system that spawns Rollback entities>) (ggrs_hip_add_spawn_system): `source` defines
        `__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload)`, `bundle` = the
        components a spawned entity has, `bindings` = [(comp, word), ..] the words the spawner writes (e.f32(i) ..); how many
        entities a frame spawns is AdvanceFrame.spawn_count, what they read AdvanceFrame.spawn_payload."""
        d = _ffi.SpawnSystemDesc()
        d.name, d.source, d.n_bindings, d.payload_stride = name.encode(), source.encode(), len(bindings), payload_stride
        d.bundle_mask = sum(1 << c for c in bundle)
        for i, (c, w) in enumerate(bindings): d.comp[i], d.word[i] = c, w
        for i, v in enumerate(iparam): d.iparam[i] = v
        for i, v in enumerate(fparam): d.fparam[i] = v
        self._check(self._lib.ggrs_hip_add_spawn_system(self._p, C.byref(d)))

    def register_component_strategy(self, comp: int, stored_word_bytes: int, stored_n_words: int, source: str):
        """ComponentSnapshotPlugin<S: Strategy> with a user-written S (ggrs_hip_register_component_strategy): snapshots hold
        stored_n_words words of stored_word_bytes; `source` defines ggrs_store(const GgrsWords& target, GgrsWords& stored) and
        ggrs_load(const GgrsWords& stored, GgrsWords& target)."""
        self._check(self._lib.ggrs_hip_register_component_strategy(self._p, comp, stored_word_bytes, stored_n_words, source.encode()))

    def host_timeline(self, enable: int = -1) -> dict:
        """ggrs_hip_host_timeline: where the host spends a tick (microseconds since the timeline was enabled); enable 1 = reset + start, 0 = stop."""
        us = (C.c_double * _ffi.TIMELINE_FIELDS)(); cnt = (C.c_uint64 * 3)()
        self._check(self._lib.ggrs_hip_host_timeline(self._p, enable, us, cnt))
        names = ["enqueue_us", "validate_us", "launch_call_us", "collect_us", "event_wait_us", "tag_wait_us", "host_fold_us"]
        out = {n: float(us[i]) for i, n in enumerate(names)}
        out.update(enqueue_calls=int(cnt[0]), collect_calls=int(cnt[1]), launches=int(cnt[2]))
        return out

    def generated_kernel_source(self, compile: bool = False, steady: bool = False) -> str:
        """The request-group kernel the library writes for this world at seal (ggrs_hip_generated_kernel_source); steady=True: the copy
        specialised for the world's steady SyncTest tick; with compile=True it is also built for gfx950 with hiprtc.  Works on a
        GGRS_WORLD_LAYOUT_ONLY world (no GPU)."""
        form = _ffi.KERNEL_FORM_STEADY if steady else _ffi.KERNEL_FORM_TILES
        need = C.c_uint64(0)
        self._check(self._lib.ggrs_hip_generated_kernel_source(self._p, form, None, 0, C.byref(need), 0))
        buf = C.create_string_buffer(need.value)
        self._check(self._lib.ggrs_hip_generated_kernel_source(self._p, form, buf, need.value, C.byref(need), 1 if compile else 0))
        return buf.value.decode()

    def checksum_component_custom(self, comp: int, source: str):
        """checksum_component::<T>(fn(&T) -> u64) with a user-written hasher (ggrs_hip_checksum_component_custom): HIP C++
        defining `__device__ ggrs_u64 ggrs_hash(const GgrsComponent& c)`."""
        self._check(self._lib.ggrs_hip_checksum_component_custom(self._p, comp, source.encode()))

    def close(self):
        if getattr(self, "_p", None):
            self._lib.ggrs_hip_world_destroy(self._p)
            self._p = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def set_synctest_check_distance(self, cd: int):
        self._check(self._lib.ggrs_hip_set_synctest_check_distance(self._p, cd))

    # ---- asynchronous request batches (include/ggrs_hip.h: enqueue / collect)
    def enqueue_requests(self, requests) -> int:
        arr, keep, n_save = self.build_requests(requests)
        self._check(self._lib.ggrs_hip_enqueue_requests(self._p, arr, len(requests), None))
        return n_save

    def enqueue_requests_raw(self, arr, n: int):
        self._check(self._lib.ggrs_hip_enqueue_requests(self._p, arr, n, None))

    def collect_checksums(self, max_saves: int = 256) -> list:
        out = (C.c_uint64 * (2 * max_saves))()
        got = C.c_uint32(0)
        self._check(self._lib.ggrs_hip_collect_checksums(self._p, out, max_saves, C.byref(got)))
        return [int(out[2 * i]) | (int(out[2 * i + 1]) << 64) for i in range(got.value)]

    def collect_checksums_raw(self, out, max_saves: int):
        self._check(self._lib.ggrs_hip_collect_checksums(self._p, out, max_saves, None))

    def pending_batches(self) -> int:
        return int(self._lib.ggrs_hip_pending_batches(self._p))

    def synchronize(self):
        self._check(self._lib.ggrs_hip_synchronize(self._p))

    def state_bytes(self) -> int:
        return int(self._lib.ggrs_hip_state_bytes(self._p))

    def live_state_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.ggrs_hip_live_state_ptr(self._p, C.byref(p)))
        return p.value

    def adopt_live_state(self):
        self._check(self._lib.ggrs_hip_adopt_live_state(self._p))

    def column_device_ptr(self, comp: int, word: int):
        """(device address of element 0, tile stride in bytes): element e of the column lives at
        ptr + (e // 8192) * tile_stride + (e % 8192) * word_bytes (tile-major word columns, 8192-slot layout tiles)."""
        p = C.c_void_p()
        ts = C.c_uint64(0)
        self._check(self._lib.ggrs_hip_column_device_ptr(self._p, comp, word, C.byref(p), C.byref(ts)))
        return p.value, ts.value

    def profile_enable(self, on: bool = True):
        self._check(self._lib.ggrs_hip_profile_enable(self._p, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * _ffi.KERNEL_CLASSES)()
        n = (C.c_uint64 * _ffi.KERNEL_CLASSES)()
        self._check(self._lib.ggrs_hip_profile_read(self._p, ms, n))
        names = ["save", "load", "advance", "checksum", "tick"]
        return {names[i]: (float(ms[i]), int(n[i])) for i in range(_ffi.KERNEL_CLASSES)}

    def profile_bytes(self):
        """Algorithmic bytes the launches of each kernel class were asked to move since profile_enable(True)."""
        b = (C.c_uint64 * _ffi.KERNEL_CLASSES)()
        self._check(self._lib.ggrs_hip_profile_read_bytes(self._p, b))
        return dict(zip(["save", "load", "advance", "checksum", "tick"], (int(x) for x in b)))

    def profile_launches(self, cls: str = "tick", cap: int = 65536):
        """Duration (us) of every launch of one kernel class since profile_enable(True), in submission order."""
        idx = ["save", "load", "advance", "checksum", "tick"].index(cls)
        buf = (C.c_float * cap)()
        n = C.c_uint32(0)
        self._check(self._lib.ggrs_hip_profile_read_launches(self._p, idx, buf, cap, C.byref(n)))
        return [float(buf[i]) for i in range(min(cap, n.value))]

    def specialise_wait(self) -> bool:
        """Block until a build of the kernel specialised for the session's steady group shape (if one is in flight) has finished;
        True when such a kernel is ready (include/ggrs_hip.h ggrs_hip_specialise_wait)."""
        rc = int(self._lib.ggrs_hip_specialise_wait(self._p))
        if rc < 0: self._check(rc)
        return rc == 1

    def kernel_info(self) -> dict:
        """ggrs_hip_world_kernel_info as a dict: arena kind, run-time compiler state, which kernel serves the world."""
        need = C.c_uint64(0)
        self._check(self._lib.ggrs_hip_world_kernel_info(self._p, None, 0, C.byref(need)))
        buf = C.create_string_buffer(int(need.value))
        self._check(self._lib.ggrs_hip_world_kernel_info(self._p, buf, need.value, None))
        return dict(l.split("=", 1) for l in buf.value.decode().splitlines() if "=" in l)
