"""Scripted fan-out sessions shared by tests/test_gpu_zfanout.py (the C-ABI path on the GPU) and tests/test_fanout_gloo.py (oracle worlds over gloo).

ADOPTION ("when the true input arrives, the matching branch's state is adopted"): a scripted sequence of TRUE inputs drives the session.  Every
step confirms one frame with its true input and speculates D frames ahead on every branch; then the true inputs of the following frames "arrive" and, when a
branch predicted a run of them, that branch's state is adopted -- the session jumps k frames without re-simulating them on the rank that owns the branch.
Whatever the partition, every rank must end in the state ONE oracle world reaches by simulating the true inputs in a straight line, and every checksum the
session observed on its way (confirmed frames, adopted frames) must be that walk's."""
from __future__ import annotations

import numpy as np

import common as cm

RATE = 50


def true_input(frame: int) -> int:
    """runs of three identical frames: spawn, spawn, spawn, idle, idle, idle, ..."""
    return cm.INPUT_SPAWN if (frame // 3) % 2 == 0 else 0


def branch_input(b: int, frame: int) -> int:
    """even branches predict the spawn key held, odd branches predict it released (frame-invariant, like repeat-last-input prediction)."""
    return cm.INPUT_SPAWN if b % 2 == 0 else 0


def build_world(w, n: int, root: bool, warm: int = 3):
    ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
    if root:
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        for _ in range(warm):
            w.advance((0,))
    else:
        w.spawn(0, {})
    return ids


def run_adopt_session(fan, n_branches_total: int, steps: int, broadcast_every: int = 0):
    """Drives `fan` (a SpeculativeFanout with retain != none, or an oracle-world one) through `steps` confirm-speculate-adopt rounds.
    Returns [(frame, checksum)] of every frame whose checksum the session observed."""
    seen = []
    D = fan.D
    for s in range(steps):
        out = fan.step()
        seen.append((out["confirmed_frame"], out["confirmed_checksum"]))
        C = fan.confirmed
        # the true inputs of frames C, C+1, .. arrive: which branches predicted a run of them, and how long?
        best_k, cands = 0, []
        for b in range(n_branches_total):
            k = 0
            while k < D - 1 and branch_input(b, C + k) == true_input(C + k): k += 1
            if k > best_k: best_k, cands = k, [b]
            elif k == best_k and k: cands.append(b)
        if best_k:
            b = cands[s % len(cands)]                                      # a different owner from round to round
            fan.adopt(b, best_k, broadcast=bool(broadcast_every and s % broadcast_every == broadcast_every - 1))
            seen.append((fan.confirmed, fan.adopted_checksum))
    return seen


def straight_line_reference(n: int, frames: int, cap: int, warm: int = 3):
    """ONE oracle world simulating the true inputs frame by frame: {frame: Checksum(u128) of SaveWorld at that frame} and a function state_at(frame)."""
    import bevy_ggrs_amd as bg
    from oracle.binding import OracleWorld
    w = OracleWorld(cap, 4)
    ids = build_world(w, n, True, warm)
    w.set_depth(2)
    fn = cm.frame_spawn_fn(RATE)
    cs = {}
    first = w.frame
    states = {}
    for f in range(first, first + frames + 1):
        w.set_confirmed(f)
        cs[f] = w.handle_requests([bg.SaveGameState(f)])[0]
        states[f] = None
        a = bg.AdvanceFrame((true_input(f),))
        if true_input(f) & cm.INPUT_SPAWN:
            a.spawn_vx, a.spawn_vy = fn(f)
        w.handle_requests([a])
    return cs, w, ids


def state_at(n: int, frame: int, cap: int, warm: int = 3):
    """The straight-line oracle world's observable state at `frame`."""
    import bevy_ggrs_amd as bg
    from oracle.binding import OracleWorld
    w = OracleWorld(cap, 4)
    ids = build_world(w, n, True, warm)
    fn = cm.frame_spawn_fn(RATE)
    while w.frame < frame:
        f = w.frame
        a = bg.AdvanceFrame((true_input(f),))
        if true_input(f) & cm.INPUT_SPAWN:
            a.spawn_vx, a.spawn_vy = fn(f)
        w.handle_requests([a])
    return cm.snapshot_state(w, ids)
