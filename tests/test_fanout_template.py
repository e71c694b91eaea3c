"""The fan-out driver's pre-marshalled request list (bevy_ggrs_amd/fanout.py `_template` / `_patch`): what it hands the library for
confirmed frame C must be, field for field and payload byte for payload byte, what marshalling `_requests(C)` from scratch gives --
with and without the shared prefix, with and without a spawn system (branches whose predicted input byte carries INPUT_SPAWN point at
the payload of the frame they advance).  CPU only: the arrays are compared, nothing is executed."""
import ctypes as C

import numpy as np
import pytest

import common as cm
from bevy_ggrs_amd.fanout import SpeculativeFanout
from oracle.binding import OracleWorld


class _Dist:
    def __init__(self, rank, size): self.rank, self.size = rank, size
    def get_rank(self): return self.rank
    def get_world_size(self): return self.size


def _fields(arr, n):
    out = []
    for i in range(n):
        q = arr[i]
        inp = bytes(q.inputs[k] for k in range(q.n_inputs))
        cnt = int(q.spawn_count)
        pay = (np.ctypeslib.as_array(q.spawn_vx, (cnt,)).tobytes(), np.ctypeslib.as_array(q.spawn_vy, (cnt,)).tobytes()) if cnt else None
        assert bool(q.spawn_vx) == bool(cnt) and bool(q.spawn_vy) == bool(cnt)
        out.append((int(q.kind), int(q.frame), int(q.dt_bits), inp, cnt, pay))
    return out


@pytest.mark.parametrize("spawn", [False, True])
@pytest.mark.parametrize("bpr,share", [(1, True), (5, True), (5, False), (32, True)])
@pytest.mark.parametrize("rank", [0, 2])
def test_patched_template_equals_fresh_marshalling(spawn, bpr, share, rank):
    D = 4
    w = OracleWorld(64, D + 2)
    cm.build_particles(w, with_spawn=spawn)
    # predictions that change from frame to frame AND differ per branch; about half of them hold the spawn key
    branch_input = lambda b, f: ((b * 7 + f * 3) & 0x0F) | (cm.INPUT_SPAWN if (b + f) % 2 else 0)
    fan = SpeculativeFanout(w, _Dist(rank, 3), D, None, branches_per_rank=bpr, branch_input=branch_input, confirmed_input=lambda f: f & 3,
                            spawn_fn=cm.frame_spawn_fn(9) if spawn else None, share_prefix=share)
    t = fan._template()
    for Cf in (0, 1, 2, 7, 8, 100, 101):
        fan._patch(t, Cf)
        want_arr, keep, n_save = w.build_requests(fan._requests(Cf))
        assert t["n_save"] == n_save == fan.saves_per_step
        got, want = _fields(t["arr"], t["n"]), _fields(want_arr, t["n"])
        assert got == want, next((i, g, x) for i, (g, x) in enumerate(zip(got, want)) if g != x)
        if spawn: assert any(f[4] for f in got) and not all(f[4] for f in got if f[0] == want[1][0])
    w.close()


def test_constant_predictions_leave_the_input_bytes_alone():
    """config 5's default predictions (the branch id, every frame): after the first step only frames and payload pointers are rewritten."""
    D = 8
    w = OracleWorld(64, D + 2)
    cm.build_particles(w, with_spawn=True)
    fan = SpeculativeFanout(w, _Dist(0, 1), D, None, branches_per_rank=16, spawn_fn=cm.frame_spawn_fn(5))
    t = fan._template()
    fan._patch(t, 0)
    first = t["inputs"]
    fan._patch(t, 1)
    assert t["inputs"] is first
    assert _fields(t["arr"], t["n"]) == _fields(w.build_requests(fan._requests(1))[0], t["n"])
    w.close()
