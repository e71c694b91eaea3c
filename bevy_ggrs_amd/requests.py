"""GgrsRequest mirror (ggrs crate; consumed by handle_requests,
/root/reference/src/schedule_systems.rs:170-289)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np


@dataclass
class SaveGameState:
    """GgrsRequest::SaveGameState { cell, frame } (schedule_systems.rs:223-237)."""
    frame: int


@dataclass
class LoadGameState:
    """GgrsRequest::LoadGameState { frame, .. } (schedule_systems.rs:238-250)."""
    frame: int


@dataclass
class AdvanceFrame:
    """GgrsRequest::AdvanceFrame { inputs } (schedule_systems.rs:251-268).

    `inputs`: PlayerInputs<T> (src/lib.rs:98) -- one T::Input per player: an int (one byte: Config<Input = u8>) or
    `input_bytes` bytes (bytes / a sequence of ints; World.set_input_layout); `status`: one InputStatus per player
    (INPUT_CONFIRMED / _PREDICTED / _DISCONNECTED), None = every input Confirmed.
    `spawn_vx/vy` carry the host-side ParticleRng draw for the PARTICLES_SPAWN system
    (examples/stress_tests/particles.rs:258-270); `spawn_count` + `spawn_payload` (bytes / a numpy array) feed a
    user-written spawn system (World.add_spawn_system); `dt_bits` = 0 derives Time::delta_secs
    from the frame number (src/time.rs:63-87)."""
    inputs: Sequence = field(default_factory=tuple)
    dt_bits: int = 0
    spawn_vx: Optional[np.ndarray] = None
    spawn_vy: Optional[np.ndarray] = None
    status: Optional[Sequence[int]] = None
    spawn_count: int = 0
    spawn_payload: Optional[object] = None
