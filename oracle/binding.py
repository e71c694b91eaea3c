"""ctypes binding of the CPU oracle (oracle/_build/libggrs_oracle.so) -- TEST INFRASTRUCTURE.

Exposes the same Python surface as bevy_ggrs_amd.World so a test can drive the HIP path and the
oracle with one script and compare.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under bevy_ggrs_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from bevy_ggrs_amd._ffi import Request, SystemDesc
from bevy_ggrs_amd.world import WorldBase

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libggrs_oracle.so")

FLAT, REFSHAPED = 0, 1
# checksum_component::<T>(fn(&T) -> u64): the component's words of one entity (n_words * word_bytes bytes), its slot, user data
HASH_FN = C.CFUNCTYPE(C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64, C.c_void_p)


class FrameView(C.Structure):
    """What a user-written system sees of the frame (oracle/ggrs_oracle.cpp FrameView): dt, frame, PlayerInputs<T> bytes + InputStatus bytes, constants."""
    _fields_ = [("dt", C.c_float), ("frame", C.c_int32), ("n_inputs", C.c_uint32), ("input_bytes", C.c_uint32),
                ("inputs", C.POINTER(C.c_uint8)), ("status", C.POINTER(C.c_uint8)), ("fparam", C.c_float * 4), ("iparam", C.c_int64 * 2)]

    def input(self, h: int) -> bytes:
        ib = self.input_bytes
        return bytes(self.inputs[h * ib:(h + 1) * ib])


CUSTOM_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(FrameView), C.POINTER(C.c_int32), C.c_void_p)
SPAWN_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(FrameView), C.POINTER(C.c_uint8), C.c_void_p)
STORE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)
LOAD_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)


def build(force: bool = False):
    src = os.path.join(_HERE, "ggrs_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    sig = {
        "gor_seahash_buffer": (C.c_uint64, [P, C.c_uint64]),
        "gor_seahash_stream": (C.c_uint64, [P, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint32]),
        "gor_diffuse": (C.c_uint64, [C.c_uint64]),
        "gor_inner_hash_units": (C.c_uint64, [C.POINTER(C.c_uint32), C.c_uint32]),
        "gor_entity_part": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "gor_finalize_part": (C.c_uint64, [C.c_uint64]),
        "gor_entity_checksum": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "gor_dt_bits": (C.c_uint32, [C.c_uint64, C.c_int32]),
        "gor_ring_create": (P, [C.c_uint64]),
        "gor_ring_destroy": (None, [P]),
        "gor_ring_set_depth": (None, [P, C.c_uint64]),
        "gor_ring_push": (None, [P, C.c_int32, C.c_uint32]),
        "gor_ring_confirm": (None, [P, C.c_int32]),
        "gor_ring_rollback": (C.c_int, [P, C.c_int32]),
        "gor_ring_get": (C.c_int, [P, C.POINTER(C.c_uint32)]),
        "gor_ring_peek": (C.c_int, [P, C.c_int32, C.POINTER(C.c_uint32)]),
        "gor_ring_len": (C.c_uint64, [P]),
        "gor_world_create": (P, [C.c_uint64, C.c_uint32, C.c_int]),
        "gor_world_destroy": (None, [P]),
        "gor_last_error": (C.c_char_p, [P]),
        "gor_register_component": (C.c_int, [P, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
        "gor_register_component_ex": (C.c_int, [P, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
        "gor_set_component_default": (C.c_int, [P, C.c_uint32, P]),
        "gor_checksum_component": (C.c_int, [P, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]),
        "gor_checksum_component_custom": (C.c_int, [P, C.c_uint32, HASH_FN, P]),
        "gor_add_system": (C.c_int, [P, C.POINTER(SystemDesc)]),
        "gor_spawn": (C.c_int, [P, C.c_uint64, C.c_uint64, C.POINTER(P), C.POINTER(C.c_uint64)]),
        "gor_despawn": (C.c_int, [P, C.c_uint64]),
        "gor_despawn_rollback": (C.c_int, [P, C.c_uint64]),
        "gor_download_disabled": (C.c_int, [P, P, C.c_uint64]),
        "gor_download_despawned_frames": (C.c_int, [P, C.c_uint64, C.c_uint64, P]),
        "gor_insert_component": (C.c_int, [P, C.c_uint32, C.c_uint64, P]),
        "gor_remove_component": (C.c_int, [P, C.c_uint32, C.c_uint64]),
        "gor_upload_word": (C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, P]),
        "gor_download_word": (C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, P]),
        "gor_download_alive": (C.c_int, [P, P, C.c_uint64]),
        "gor_download_present": (C.c_int, [P, C.c_uint32, P, C.c_uint64]),
        "gor_len": (C.c_uint64, [P]),
        "gor_active_count": (C.c_uint64, [P]),
        "gor_frame": (C.c_int32, [P]),
        "gor_set_frame": (None, [P, C.c_int32]),
        "gor_set_frame_rate": (None, [P, C.c_uint64]),
        "gor_set_depth": (None, [P, C.c_uint32]),
        "gor_set_confirmed": (None, [P, C.c_int, C.c_int32]),
        "gor_has_snapshot": (C.c_int, [P, C.c_int32]),
        "gor_snapshot_count": (C.c_uint64, [P]),
        "gor_save": (C.c_int, [P, C.POINTER(C.c_uint64)]),
        "gor_load": (C.c_int, [P, C.c_int32]),
        "gor_advance": (C.c_int, [P, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint64,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "gor_handle_requests": (C.c_int, [P, C.POINTER(Request), C.c_uint32, C.POINTER(C.c_uint64)]),
        "gor_set_input_layout": (C.c_int, [P, C.c_uint32, C.c_uint32]),
        "gor_add_custom_system": (C.c_int, [P, CUSTOM_FN, P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
        "gor_add_spawn_system": (C.c_int, [P, SPAWN_FN, P, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
        "gor_register_component_strategy": (C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint32, STORE_FN, LOAD_FN, P]),
        "gor_bench_synctest": (C.c_double, [P, C.c_uint32, C.c_uint32, C.c_uint32]),
        "gor_replay_synctest": (C.c_double, [P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]),
        "gor_replay_synctest_from": (C.c_double, [P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]),
        "gor_set_ref_component_threads": (None, [C.c_int]),
        "gor_num_threads": (C.c_int, []),
        "gor_set_num_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
# the GPU box has 256 host cores; a 256-thread OpenMP team per tiny parallel region makes the
# oracle crawl, so the checker defaults to a small team (bench.py sets its own count explicitly)
lib.gor_set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))


class OracleWorld(WorldBase):
    _lib = lib
    _prefix = "gor_"

    def __init__(self, capacity: int, max_depth: int = 8, mode: int = FLAT):
        self._p = C.c_void_p(lib.gor_world_create(capacity, max_depth, mode))
        self._comps = []
        self.capacity = capacity
        self.mode = mode

    def close(self):
        if getattr(self, "_p", None):
            lib.gor_world_destroy(self._p)
            self._p = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def active_count(self) -> int:
        return int(lib.gor_active_count(self._p))

    def checksum_component_custom(self, comp: int, fn):
        """checksum_component::<T>(hasher) with an arbitrary hasher: fn(words: bytes, slot: int) -> int, `words` = the
        component's n_words * word_bytes bytes of one entity.  (A Python callback per entity: small worlds only.)"""
        _, wb, nw = self._comps[comp]
        nbytes = wb * nw
        cb = HASH_FN(lambda p, slot, _u: int(fn(bytes(p[:nbytes]), int(slot))) & 0xFFFFFFFFFFFFFFFF)
        self._keep = getattr(self, "_keep", []) + [cb]
        self._check(lib.gor_checksum_component_custom(self._p, comp, cb, None))

    # ---- user-written systems / spawners / strategies as Python callbacks (one call per entity: small worlds only)
    def _params(self, iparam, fparam):
        ip = (C.c_int64 * 2)(*(list(iparam) + [0, 0])[:2]); fp = (C.c_float * 4)(*(list(fparam) + [0, 0, 0, 0])[:4])
        return ip, fp

    def add_custom_system(self, fn, bindings, iparam=(), fparam=(), name: str = "custom"):
        """fn(words: list[int], slot: int, frame: FrameView) -> (new words, kill[, n_spawn]) with kill 0 / 1 = despawn() / 2 = despawn_rollback() and n_spawn =
        e.spawn(n): how many Rollback entities this call spawns (a spawn system with payload_stride 0xFFFFFFFF builds them from this call's words)."""
        n = len(bindings)

        def thunk(words, slot, f, kill, _u):
            r = fn([int(words[i]) for i in range(n)], int(slot), f.contents)
            new, k, ns = (r[0], r[1], r[2]) if len(r) > 2 else (r[0], r[1], 0)
            for i in range(n): words[i] = int(new[i]) & 0xFFFFFFFFFFFFFFFF
            kill[0] = int(k) | (min(max(int(ns), 0), 255) << 8)
        cb = CUSTOM_FN(thunk)
        self._keep = getattr(self, "_keep", []) + [cb]
        comp = (C.c_uint32 * n)(*[c for c, _ in bindings]); word = (C.c_uint32 * n)(*[w for _, w in bindings])
        ip, fp = self._params(iparam, fparam)
        self._check(lib.gor_add_custom_system(self._p, cb, None, n, comp, word, ip, fp))

    def add_spawn_system(self, fn, bundle, bindings=(), payload_stride: int = 0, iparam=(), fparam=(), name: str = "spawn"):
        """fn(words: list[int], slot: int, k: int, frame: FrameView, payload: ctypes pointer to the entity's record) -> new words."""
        n = len(bindings)

        def thunk(words, slot, k, f, payload, _u):
            new = fn([int(words[i]) for i in range(n)], int(slot), int(k), f.contents, payload)
            for i in range(n): words[i] = int(new[i]) & 0xFFFFFFFFFFFFFFFF
        cb = SPAWN_FN(thunk)
        self._keep = getattr(self, "_keep", []) + [cb]
        comp = (C.c_uint32 * max(1, n))(*[c for c, _ in bindings]); word = (C.c_uint32 * max(1, n))(*[w for _, w in bindings])
        ip, fp = self._params(iparam, fparam)
        self._check(lib.gor_add_spawn_system(self._p, cb, None, sum(1 << c for c in bundle), payload_stride, n, comp, word, ip, fp))

    def register_component_strategy(self, comp: int, stored_word_bytes: int, stored_n_words: int, store, load):
        """store(target words: list[int]) -> stored words; load(stored words: list[int]) -> target words (Strategy::store / ::load)."""
        _, _wb, nw = self._comps[comp]

        def st(tg, out, _u):
            r = store([int(tg[i]) for i in range(nw)])
            for i in range(stored_n_words): out[i] = int(r[i]) & 0xFFFFFFFFFFFFFFFF

        def ld(sv, out, _u):
            r = load([int(sv[i]) for i in range(stored_n_words)])
            for i in range(nw): out[i] = int(r[i]) & 0xFFFFFFFFFFFFFFFF
        cbs, cbl = STORE_FN(st), LOAD_FN(ld)
        self._keep = getattr(self, "_keep", []) + [cbs, cbl]
        self._check(lib.gor_register_component_strategy(self._p, comp, stored_word_bytes, stored_n_words, cbs, cbl, None))

    @staticmethod
    def sea_hasher():
        """A SeaHasher (checksum_hasher(), snapshot/mod.rs:318-320) for Python-side custom hashers: .write(bytes), .finish()."""
        class _H:
            def __init__(self): self.buf = b""
            def write(self, b): self.buf += bytes(b); return self
            def finish(self):
                a = (C.c_uint8 * max(1, len(self.buf))).from_buffer_copy(self.buf or b"\0")
                sizes = (C.c_uint32 * 1)(len(self.buf))
                return int(lib.gor_seahash_stream(a, len(self.buf), sizes, 1))
        return _H()

    def set_synctest_check_distance(self, cd: int):
        self._cd = cd   # the oracle applies the rule in Python (see tests/common.py)

    def bench_synctest(self, d: int, warm_ticks: int, ticks: int) -> float:
        return float(lib.gor_bench_synctest(self._p, d, warm_ticks, ticks))

    def replay_synctest(self, d: int, ticks: int, timed_ticks: int):
        """`ticks` SyncTest ticks from the current frame; returns (seconds of the last `timed_ticks`, [checksum per Save])."""
        cap = (d + 1) * ticks + 8
        buf = (C.c_uint64 * (2 * cap))()
        n = C.c_uint64(0)
        secs = float(lib.gor_replay_synctest(self._p, d, ticks, timed_ticks, buf, cap, C.byref(n)))
        return secs, [int(buf[2 * k]) | (int(buf[2 * k + 1]) << 64) for k in range(n.value)]


    def replay_synctest_from(self, d: int, skip_frames: int, ticks: int):
        """Fast-forward `skip_frames` advance-only frames + d+1 plain ticks, then `ticks` SyncTest ticks;
        returns (seconds of those ticks, [checksum per Save of those ticks])."""
        cap = (d + 1) * ticks + 8
        buf = (C.c_uint64 * (2 * cap))()
        n = C.c_uint64(0)
        secs = float(lib.gor_replay_synctest_from(self._p, d, skip_frames, ticks, buf, cap, C.byref(n)))
        return secs, [int(buf[2 * k]) | (int(buf[2 * k + 1]) << 64) for k in range(n.value)]


class OracleRing:
    """GgrsSnapshots<u32, u32> (src/snapshot/mod.rs:357) for the 11 known-answer tests."""

    def __init__(self, depth: int):
        self._p = C.c_void_p(lib.gor_ring_create(depth))

    def __del__(self):
        if self._p: lib.gor_ring_destroy(self._p); self._p = None

    def push(self, frame, v): lib.gor_ring_push(self._p, frame, v)
    def confirm(self, frame): lib.gor_ring_confirm(self._p, frame)

    def rollback(self, frame):
        if lib.gor_ring_rollback(self._p, frame) != 0:
            raise RuntimeError(f"Could not rollback to {frame}: no snapshot at that moment could be found.")

    def get(self):
        out = C.c_uint32(0)
        if lib.gor_ring_get(self._p, C.byref(out)) != 0: raise RuntimeError("no snapshot available")
        return out.value

    def peek(self, frame):
        out = C.c_uint32(0)
        return out.value if lib.gor_ring_peek(self._p, frame, C.byref(out)) == 0 else None

    def __len__(self): return int(lib.gor_ring_len(self._p))
