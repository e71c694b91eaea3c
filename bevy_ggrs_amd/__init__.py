"""bevy_ggrs_amd -- MI355X-native rollback re-simulation engine behind bevy_ggrs's API names.

Only the snapshot-and-resimulate hot path lives here: registered ECS
component columns as SoA arrays in HBM, snapshot save/restore, the deterministic GgrsSchedule
step and the SeaHash checksum as hand-written gfx950 kernels (csrc/), driven through the C ABI
of libggrs_hip.so (include/ggrs_hip.h).
"""
from ._ffi import (COMP_NO_ROLLBACK, COMP_ROLLBACK, DESPAWN_IMMEDIATE, DESPAWN_ROLLBACK, GGRS_E_CAPACITY, GGRS_E_INVALID, GGRS_E_NO_DEVICE, GGRS_E_NO_SNAPSHOT,  # noqa: F401
                   GGRS_WORLD_LAYOUT_ONLY, GGRS_WORLD_NO_GROUPS, GGRS_WORLD_NT_COPY, GGRS_WORLD_UNFUSED, GgrsHipError, SYS_ADD_U32, SYS_BOX_MOVE, SYS_PARTICLES_SPAWN,
                   SYS_PARTICLES_UPDATE, SYS_SAT_SUB_DESPAWN, SYS_TTL_DESPAWN, INPUT_CONFIRMED, INPUT_PREDICTED, INPUT_DISCONNECTED)
from .requests import AdvanceFrame, LoadGameState, SaveGameState  # noqa: F401
from .world import World  # noqa: F401
