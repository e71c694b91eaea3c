"""Speculative re-simulation fan-out across the GPUs of one node (BASELINE.json config 5).

The reference has no multi-process story: one timeline's D resimulated
frames are a sequential chain.  What shards naturally is *speculation*: B predicted-input
branches, all starting from the same confirmed-frame snapshot, are independent.  One process
per GPU (torch.distributed, backend "nccl" == RCCL over xGMI):

  sync_confirmed()   rank `src` broadcasts its packed live state block (header + liveness and
                     presence masks + every registered word column; one contiguous buffer,
                     include/ggrs_hip.h `ggrs_hip_live_state_ptr`) -- ONE RCCL broadcast.  xGMI is
                     point-to-point, so this costs ~state_bytes / per-link bandwidth; it is done at
                     start-up and after a detected desync, NOT every step:
  step()             every rank re-simulates its own branches for D frames from its replica of
                     the confirmed snapshot [Load(C), (Advance(b_i), Save) x D per branch], then
                     applies the *confirmed* input to its replica [Load(C), Advance(c), Save(C+1)]
                     -- rollback netcode's determinism keeps the replicas bit-identical, so moving
                     the 1-byte input replaces re-broadcasting 60 B/entity -- and ONE all-gather
                     carries every branch's per-frame Checksum(u128) plus the replica's confirmed
                     checksum, which doubles as cross-rank desync detection.

No data-path collective touches entity columns inside step(); per-GPU work is fixed as the
world size grows (weak scaling).  `exchange` abstracts where the packed state lives so the same
control flow runs on gloo/CPU worlds in tests/test_fanout_gloo.py.
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


class HipStateExchange:
    """Packed-state broadcast for a `bevy_ggrs_amd.World` whose arena is a torch CUDA tensor:
    the live state block is arena[0:state_bytes], so RCCL reads/writes HBM in place."""

    def __init__(self, world, arena):
        self.world, self.arena = world, arena

    def broadcast(self, dist, src: int):
        w = self.world
        nbytes = w.state_bytes()
        ptr = w.live_state_ptr()            # refreshes the header (len, frame) on every rank
        assert ptr == self.arena.data_ptr(), "live block must be the head of the torch arena"
        dist.broadcast(self.arena[:nbytes], src=src)
        import torch
        torch.cuda.current_stream().synchronize()
        w.adopt_live_state()

    def all_gather_u64(self, dist, values: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(values.view(np.int64)).to(self.arena.device)
        out = torch.empty((dist.get_world_size(), t.numel()), dtype=torch.int64, device=self.arena.device)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().view(np.uint64)


def make_torch_world(bg, capacity: int, max_depth: int, n_components: int, bytes_per_slot: int,
                     device, flags: int = 0):
    """World whose device memory is ONE torch uint8 tensor (so collectives can address it)."""
    import torch
    from . import _ffi
    nbytes = int(_ffi.lib.ggrs_hip_arena_bytes(capacity, max_depth, n_components, bytes_per_slot))
    arena = torch.empty(nbytes, dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    dev_index = device.index if hasattr(device, "index") and device.index is not None else 0
    w = bg.World(capacity, max_depth=max_depth, device=dev_index, stream=stream,
                 arena_ptr=arena.data_ptr(), arena_bytes=nbytes, flags=flags)
    return w, arena


def default_branch_input(branch: int, frame: int) -> int:
    """256 distinct predicted-input sequences: the branch id repeated every frame."""
    return branch & 0xFF


class SpeculativeFanout:
    def __init__(self, world, dist, depth: int, exchange, branches_per_rank: int = 1,
                 branch_input: Callable[[int, int], int] = default_branch_input,
                 confirmed_input: Callable[[int], int] = lambda frame: 0,
                 spawn_fn: Optional[Callable[[int], tuple]] = None, spawn_mask: int = 1 << 4,
                 num_players: int = 1):
        self.w, self.dist, self.D, self.x = world, dist, depth, exchange
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.bpr = branches_per_rank
        self.branch_input, self.confirmed_input = branch_input, confirmed_input
        self.spawn_fn, self.spawn_mask = spawn_fn, spawn_mask
        self.num_players = num_players
        self.synced = False
        self.last: dict = {}
        world.set_depth(depth + 1)

    # ------------------------------------------------------------------ helpers
    def branch_ids(self, rank: Optional[int] = None) -> List[int]:
        r = self.rank if rank is None else rank
        return [r * self.bpr + j for j in range(self.bpr)]      # contiguous block per rank

    def _advance(self, frame: int, inp: int) -> AdvanceFrame:
        a = AdvanceFrame((inp,) * self.num_players)
        if self.spawn_fn is not None and (inp & self.spawn_mask):
            a.spawn_vx, a.spawn_vy = self.spawn_fn(frame)      # pure function of the frame (rollback RNG)
        return a

    def sync_confirmed(self, src: int = 0):
        """Broadcast `src`'s live world as the confirmed frame C and snapshot it on every rank."""
        self.x.broadcast(self.dist, src)
        self.w.set_confirmed(self.w.frame)
        cs = self.w.handle_requests([SaveGameState(self.w.frame)])[0]
        self.synced = True
        return cs

    # ------------------------------------------------------------------ one confirmed frame
    def step(self) -> dict:
        if not self.synced:
            self.sync_confirmed(0)
        w, D = self.w, self.D
        C = w.frame if not self.last else self.last["confirmed_frame"]
        reqs: list = []
        for b in self.branch_ids():
            reqs.append(LoadGameState(C))
            for i in range(D):
                reqs.append(self._advance(C + i, self.branch_input(b, C + i)))
                reqs.append(SaveGameState(C + i + 1))
        c_in = self.confirmed_input(C)
        last_b = self.branch_ids()[-1]
        adopted = self.bpr > 0 and D > 0 and self.branch_input(last_b, C) == c_in
        n_spec = self.bpr * D
        if adopted:
            # the last branch's frame C+1 was simulated with the true input: adopt its snapshot
            reqs.append(LoadGameState(C + 1))
        else:
            reqs += [LoadGameState(C), self._advance(C, c_in), SaveGameState(C + 1)]
        cs = w.handle_requests(reqs)
        confirmed_cs = cs[(self.bpr - 1) * D] if adopted else cs[n_spec]
        # ---- ONE all-gather: [bpr x D speculative checksums | confirmed checksum], u128 as 2 x u64
        mine = np.zeros((n_spec + 1, 2), dtype=np.uint64)
        for k in range(n_spec):
            mine[k] = (cs[k] & 0xFFFFFFFFFFFFFFFF, cs[k] >> 64)
        mine[n_spec] = (confirmed_cs & 0xFFFFFFFFFFFFFFFF, confirmed_cs >> 64)
        allv = self.x.all_gather_u64(self.dist, mine.reshape(-1)).reshape(self.size, n_spec + 1, 2)
        to_int = lambda p: int(p[0]) | (int(p[1]) << 64)
        confirmed_all = [to_int(allv[r, n_spec]) for r in range(self.size)]
        if len(set(confirmed_all)) != 1:
            self.synced = False                              # caller may sync_confirmed() again
            raise DesyncDetected(C + 1, confirmed_all)
        w.set_confirmed(C + 1)                               # discard_old_snapshots bound
        self.last = {
            "confirmed_frame": C + 1, "confirmed_checksum": confirmed_all[0], "adopted": adopted,
            "branch_checksums": {r * self.bpr + j: [to_int(allv[r, j * D + i]) for i in range(D)]
                                 for r in range(self.size) for j in range(self.bpr)},
        }
        return self.last
