"""ctypes binding of libggrs_hip.so -- the C ABI declared in include/ggrs_hip.h.

The product path has NO CPU fallback: if the HIP extension is missing this module raises at
import time, and world creation raises when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libggrs_hip.so")

GGRS_OK = 0
GGRS_E_INVALID = -1
GGRS_E_NO_SNAPSHOT = -2
GGRS_E_CAPACITY = -3
GGRS_E_HIP = -4
GGRS_E_NO_DEVICE = -5

GGRS_WORLD_DEFAULT = 0
GGRS_WORLD_UNFUSED = 2
GGRS_WORLD_NT_COPY = 4
GGRS_WORLD_NO_GROUPS = 8
GGRS_WORLD_LAYOUT_ONLY = 16

SYS_PARTICLES_UPDATE = 1
SYS_TTL_DESPAWN = 2
SYS_PARTICLES_SPAWN = 3
SYS_ADD_U32 = 4
SYS_SAT_SUB_DESPAWN = 5
SYS_BOX_MOVE = 6
SYS_CUSTOM = 7
SYS_SPAWN_CUSTOM = 8

INPUT_CONFIRMED, INPUT_PREDICTED, INPUT_DISCONNECTED = 0, 1, 2
MAX_INPUT_BYTES, MAX_PLAYERS = 16, 16
TIMELINE_FIELDS = 7

COMP_ROLLBACK, COMP_NO_ROLLBACK = 0, 1
DESPAWN_IMMEDIATE, DESPAWN_ROLLBACK = 0, 1

REQ_SAVE, REQ_LOAD, REQ_ADVANCE = 1, 2, 3

FANOUT_ID_BYTES = 128

KERNEL_SAVE, KERNEL_LOAD, KERNEL_ADVANCE, KERNEL_CHECKSUM, KERNEL_TICK, KERNEL_CLASSES = 0, 1, 2, 3, 4, 5


class WorldDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_depth", C.c_uint32), ("capacity", C.c_uint64),
                ("stream", C.c_void_p), ("arena", C.c_void_p), ("arena_bytes", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class SystemDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("comp", C.c_uint32 * 4), ("word", C.c_uint32 * 4),
                ("iparam", C.c_int64 * 2), ("fparam", C.c_float * 4)]


CUSTOM_MAX_BINDINGS = 8


class CustomSystemDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("source", C.c_char_p), ("n_bindings", C.c_uint32),
                ("comp", C.c_uint32 * CUSTOM_MAX_BINDINGS), ("word", C.c_uint32 * CUSTOM_MAX_BINDINGS),
                ("iparam", C.c_int64 * 2), ("fparam", C.c_float * 4)]


class SpawnSystemDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("source", C.c_char_p), ("bundle_mask", C.c_uint64), ("payload_stride", C.c_uint32), ("n_bindings", C.c_uint32),
                ("comp", C.c_uint32 * CUSTOM_MAX_BINDINGS), ("word", C.c_uint32 * CUSTOM_MAX_BINDINGS),
                ("iparam", C.c_int64 * 2), ("fparam", C.c_float * 4)]


class Request(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("frame", C.c_int32), ("dt_bits", C.c_uint32),
                ("n_inputs", C.c_uint32), ("inputs", C.POINTER(C.c_uint8)), ("status", C.POINTER(C.c_uint8)),
                ("spawn_count", C.c_uint64), ("spawn_vx", C.POINTER(C.c_float)),
                ("spawn_vy", C.POINTER(C.c_float)), ("spawn_payload", C.c_void_p), ("spawn_payload_bytes", C.c_uint64)]


class BranchSpawn(C.Structure):
    _fields_ = [("count", C.c_uint64), ("vx", C.c_void_p), ("vy", C.c_void_p), ("payload", C.c_void_p), ("payload_bytes", C.c_uint64)]


class BranchStep(C.Structure):
    _fields_ = [("prefix", C.POINTER(Request)), ("n_prefix", C.c_uint32), ("n_branches", C.c_uint32), ("n_frames", C.c_uint32), ("n_inputs", C.c_uint32),
                ("flags", C.c_uint32), ("n_spawn_table", C.c_uint32), ("inputs", C.c_void_p), ("status", C.c_void_p),
                ("spawn_table", C.POINTER(BranchSpawn)), ("spawn_sel", C.c_void_p)]


BRANCH_SAVE_LAST, BRANCH_RETAIN_NEWEST, BRANCH_RETAIN_ALL = 1, 2, 4
ADOPT_RECOMPUTE, ADOPT_BROADCAST = 0, 1

KERNEL_FORM_TILES, KERNEL_FORM_STEADY = 1, 3

# every symbol include/ggrs_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "ggrs_hip_abi_version": (C.c_int, []),
    "ggrs_hip_device_count": (C.c_int, []),
    "ggrs_hip_world_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "ggrs_hip_world_create_ex": (C.c_int, [C.POINTER(WorldDesc), C.POINTER(_P)]),
    "ggrs_hip_arena_bytes": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ggrs_hip_world_destroy": (None, [_P]),
    "ggrs_hip_last_error": (C.c_char_p, [_P]),
    "ggrs_hip_register_component": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_register_component_ex": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_set_component_default": (C.c_int, [_P, C.c_uint32, _P]),
    "ggrs_hip_checksum_component": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]),
    "ggrs_hip_checksum_component_custom": (C.c_int, [_P, C.c_uint32, C.c_char_p]),
    "ggrs_hip_add_system": (C.c_int, [_P, C.POINTER(SystemDesc)]),
    "ggrs_hip_add_custom_system": (C.c_int, [_P, C.POINTER(CustomSystemDesc)]),
    "ggrs_hip_register_component_strategy": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p]),
    "ggrs_hip_set_input_layout": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "ggrs_hip_add_spawn_system": (C.c_int, [_P, C.POINTER(SpawnSystemDesc)]),
    "ggrs_hip_generated_kernel_source": (C.c_int, [_P, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
    "ggrs_hip_aot_object_name": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64]),
    "ggrs_hip_host_timeline": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "ggrs_hip_set_frame_rate": (C.c_int, [_P, C.c_uint64]),
    "ggrs_hip_spawn": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "ggrs_hip_despawn": (C.c_int, [_P, C.c_uint64]),
    "ggrs_hip_despawn_rollback": (C.c_int, [_P, C.c_uint64]),
    "ggrs_hip_download_disabled": (C.c_int, [_P, _P, C.c_uint64]),
    "ggrs_hip_download_despawned_frames": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    "ggrs_hip_insert_component": (C.c_int, [_P, C.c_uint32, C.c_uint64, _P]),
    "ggrs_hip_remove_component": (C.c_int, [_P, C.c_uint32, C.c_uint64]),
    "ggrs_hip_upload_word": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _P]),
    "ggrs_hip_download_word": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _P]),
    "ggrs_hip_download_alive": (C.c_int, [_P, _P, C.c_uint64]),
    "ggrs_hip_download_present": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64]),
    "ggrs_hip_column_device_ptr": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "ggrs_hip_len": (C.c_uint64, [_P]),
    "ggrs_hip_active_count": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ggrs_hip_frame": (C.c_int32, [_P]),
    "ggrs_hip_set_frame": (C.c_int, [_P, C.c_int32]),
    "ggrs_hip_set_depth": (C.c_int, [_P, C.c_uint32]),
    "ggrs_hip_set_confirmed": (C.c_int, [_P, C.c_int, C.c_int32]),
    "ggrs_hip_has_snapshot": (C.c_int, [_P, C.c_int32]),
    "ggrs_hip_snapshot_count": (C.c_uint64, [_P]),
    "ggrs_hip_save": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ggrs_hip_load": (C.c_int, [_P, C.c_int32]),
    "ggrs_hip_advance": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint64,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ggrs_hip_handle_requests": (C.c_int, [_P, C.POINTER(Request), C.c_uint32, C.POINTER(C.c_uint64)]),
    "ggrs_hip_enqueue_requests": (C.c_int, [_P, C.POINTER(Request), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_collect_checksums": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_pending_batches": (C.c_uint32, [_P]),
    "ggrs_hip_set_synctest_check_distance": (C.c_int, [_P, C.c_int32]),
    "ggrs_hip_synchronize": (C.c_int, [_P]),
    "ggrs_hip_state_bytes": (C.c_uint64, [_P]),
    "ggrs_hip_live_state_ptr": (C.c_int, [_P, C.POINTER(_P)]),
    "ggrs_hip_adopt_live_state": (C.c_int, [_P]),
    "ggrs_hip_fanout_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "ggrs_hip_fanout_init": (C.c_int, [_P, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(_P)]),
    "ggrs_hip_fanout_sync_confirmed": (C.c_int, [_P, C.c_int]),
    "ggrs_hip_fanout_step": (C.c_int, [_P, C.POINTER(Request), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_fanout_set_interval": (C.c_int, [_P, C.c_uint32]),
    "ggrs_hip_fanout_collect": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ggrs_hip_fanout_destroy": (None, [_P]),
    "ggrs_hip_fanout_last_error": (C.c_char_p, [_P]),
    "ggrs_hip_fanout_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ggrs_hip_fanout_step_branches": (C.c_int, [_P, C.POINTER(BranchStep), C.POINTER(C.c_uint32)]),
    "ggrs_hip_fanout_adopt": (C.c_int, [_P, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(Request), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ggrs_hip_profile_enable": (C.c_int, [_P, C.c_int]),
    "ggrs_hip_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "ggrs_hip_profile_read_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ggrs_hip_profile_read_launches": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ggrs_hip_world_kernel_info": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ggrs_hip_specialise_wait": (C.c_int, [_P]),
}


def _preload_hip_runtime():
    """ONE HIP runtime per process.  torch wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, referenced by torch as plain `libamdhip64.so`); libggrs_hip.so needs
    `libamdhip64.so.7`.  If our library is loaded before torch, the loader would map /opt/rocm's copy
    for us and torch's copy for torch -- two runtimes, and the second one to initialise sees no
    device.  Mapping torch's copy first (when torch is installed; it is not imported here) makes both
    resolve to the same file regardless of import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C bevy_ggrs_amd/csrc`). "
            "bevy_ggrs_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load_library()


class GgrsHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ggrs_hip error {code}: {msg}")
        self.code = code
