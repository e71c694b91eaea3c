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

    `inputs` is one input byte per player (InputStatus is not consumed on this path).
    `spawn_vx/vy` carry the host-side ParticleRng draw for the PARTICLES_SPAWN system
    (examples/stress_tests/particles.rs:258-270); `dt_bits` = 0 derives Time::delta_secs
    from the frame number (src/time.rs:63-87)."""
    inputs: Sequence[int] = field(default_factory=tuple)
    dt_bits: int = 0
    spawn_vx: Optional[np.ndarray] = None
    spawn_vy: Optional[np.ndarray] = None
