"""SyncTest session: a restatement of ggrs's `SyncTestSession::advance_frame`.

ggrs is an un-vendored git dependency of the reference (Cargo.toml:23), so its request
sequencing is restated here from its published algorithm (parity-unpinned by reference
vectors; the reference tests that constrain it are tests/synctest.rs:84-153 and
tests/component_rollback.rs).  This is host-side control plane: it only decides WHICH
SaveGameState / LoadGameState / AdvanceFrame requests run; the requests themselves execute on
the GPU through `World.handle_requests`.

Per tick at current frame F with check distance d (after warm-up, F > d):
    [Load(F-d), Adv, (Save(F-d+i), Adv) for i in 1..d-1, Save(F), Adv]
i.e. 1 load, d saves, d+1 advances.  The first recorded checksum of each frame is kept and
every later re-save of that frame must reproduce it, otherwise MismatchedChecksum.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

from .requests import AdvanceFrame, LoadGameState, SaveGameState


class MismatchedChecksum(Exception):
    """GgrsError::MismatchedChecksum { current_frame, mismatched_frames } ->
    SyncTestMismatch event (src/lib.rs:133-139, schedule_systems.rs:104-115)."""

    def __init__(self, current_frame: int, mismatched_frames: List[int]):
        super().__init__(f"Detected checksum mismatch during rollback on frame {current_frame}, "
                         f"mismatched frames: {mismatched_frames}")
        self.current_frame = current_frame
        self.mismatched_frames = mismatched_frames


@dataclass
class _Cell:
    frame: int = -1
    checksum: Optional[int] = None


class SyncTestSession:
    """SessionBuilder::start_synctest_session equivalent."""

    def __init__(self, num_players: int = 1, check_distance: int = 2, max_prediction: int = 8,
                 input_delay: int = 0, allow_deep_check: bool = False):
        # ggrs rejects check_distance >= max_prediction; the bench harness may lift that to
        # run BASELINE's "depth 8" directly.
        if check_distance >= max_prediction and not allow_deep_check:
            raise ValueError("Check distance too big.")
        self.num_players = num_players
        self.check_distance = check_distance
        self._max_prediction = max_prediction
        self.input_delay = input_delay
        self.current_frame = 0
        self._local_inputs: Dict[int, int] = {}
        self._input_history: Dict[int, List[int]] = {}     # frame -> inputs actually used
        self._pending: Dict[int, List[int]] = {}            # frame -> inputs scheduled (delay)
        self._cells = [_Cell() for _ in range(max(max_prediction, check_distance) + 2)]
        self._checksum_history: Dict[int, Optional[int]] = {}
        self._last_saves: List[SaveGameState] = []

    # -- API used by run_synctest (schedule_systems.rs:85-118)
    def max_prediction(self) -> int:
        return self._max_prediction

    def add_local_input(self, handle: int, value: int):
        if not (0 <= handle < self.num_players):
            raise ValueError("invalid player handle")
        self._local_inputs[handle] = value

    def _inputs_for(self, frame: int) -> List[int]:
        if frame not in self._input_history:
            self._input_history[frame] = self._pending.get(frame, [0] * self.num_players)
        return self._input_history[frame]

    def _save_request(self) -> SaveGameState:
        r = SaveGameState(self.current_frame)
        self._last_saves.append(r)
        return r

    def _checksums_consistent(self, frame_to_check: int) -> bool:
        oldest = self.current_frame - self.check_distance
        for k in [k for k in self._checksum_history if k < oldest]:
            del self._checksum_history[k]
        cell = self._cells[frame_to_check % len(self._cells)]
        if cell.frame != frame_to_check:
            return True
        if cell.frame in self._checksum_history:
            return self._checksum_history[cell.frame] == cell.checksum
        self._checksum_history[cell.frame] = cell.checksum
        return True

    def advance_frame(self) -> list:
        requests: list = []
        self._last_saves = []
        cur = self.current_frame
        d = self.check_distance
        if d > 0 and cur > d:
            mismatched = [f for f in range(cur - d, cur + 1) if not self._checksums_consistent(f)]
            if mismatched:
                raise MismatchedChecksum(cur, mismatched)
            # adjust_gamestate: roll back d frames and resimulate
            frame_to = cur - d
            requests.append(LoadGameState(frame_to))
            self.current_frame = frame_to
            for i in range(d):
                inputs = self._inputs_for(self.current_frame)
                if i > 0:
                    requests.append(self._save_request())
                self.current_frame += 1
                requests.append(AdvanceFrame(tuple(inputs)))
            assert self.current_frame == cur
        if len(self._local_inputs) != self.num_players:
            raise ValueError("Missing local input while calling advance_frame().")
        vals = [self._local_inputs[h] for h in range(self.num_players)]
        self._pending[self.current_frame + self.input_delay] = vals
        self._local_inputs = {}
        if d > 0:
            requests.append(self._save_request())
        inputs = self._inputs_for(self.current_frame)
        requests.append(AdvanceFrame(tuple(inputs)))
        self.current_frame += 1
        # drop input history that can no longer be resimulated
        for k in [k for k in self._input_history if k < self.current_frame - d - 2]:
            self._input_history.pop(k, None); self._pending.pop(k, None)
        return requests

    def record_checksums(self, checksums: Sequence[int]):
        """cell.save(frame, None, Some(checksum)) for each SaveGameState of the last
        advance_frame(), in order (schedule_systems.rs:231-236)."""
        assert len(checksums) == len(self._last_saves)
        for r, cs in zip(self._last_saves, checksums):
            c = self._cells[r.frame % len(self._cells)]
            c.frame, c.checksum = r.frame, cs
