"""RCCL leg of the speculative fan-out on the one GPU a gpurun box has: world_size 1 over the
nccl backend exercises the torch-arena world, the in-place broadcast of the packed state
block, adopt_live_state and the all-gather; results must equal the oracle's serial walk
(tests/test_fanout_gloo.py covers world_size 2 control flow on CPU)."""
import os

import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm

pytestmark = pytest.mark.gpu


def test_state_block_roundtrip_through_torch_arena_and_nccl():
    import torch
    import torch.distributed as dist
    from bevy_ggrs_amd.fanout import HipStateExchange, SpeculativeFanout, make_torch_world
    from test_fanout_gloo import _serial_reference

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 50))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, D, steps, bpr = 700, 4, 6, 2
        cap = n + 100 * (steps + D + 2) * 2
        w, arena = make_torch_world(bg, cap, D + 2, 3, 60, torch.device("cuda", 0))
        ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        for _ in range(3):
            w.advance((0,))
        fan = SpeculativeFanout(w, dist, D, HipStateExchange(w, arena), branches_per_rank=bpr,
                                branch_input=lambda b, f: cm.INPUT_SPAWN if b % 2 == 0 else 0,
                                confirmed_input=lambda f: cm.INPUT_SPAWN if f % 2 == 1 else 0,
                                spawn_fn=cm.frame_spawn_fn(50))
        out = [fan.step() for _ in range(steps)]
        ref, ref_state = _serial_reference(n, D, bpr, steps)
        for got, want in zip(out, ref):
            assert got["confirmed_frame"] == want["confirmed_frame"]
            assert got["confirmed_checksum"] == want["confirmed_checksum"]
            assert got["branch_checksums"] == want["branch_checksums"]
        cm.assert_states_equal(cm.snapshot_state(w, ids), ref_state, "fanout gpu")

        # a second world adopts the first one's packed state block byte-for-byte (what a
        # receiving rank does after the broadcast)
        w2, arena2 = make_torch_world(bg, cap, D + 2, 3, 60, torch.device("cuda", 0))
        ids2 = cm.build_particles(w2, with_spawn=True, ttl_init=25)
        w2.spawn(0, {})
        nb = w.state_bytes()
        assert nb == w2.state_bytes()
        w.live_state_ptr()
        arena2[:nb].copy_(arena[:nb])
        torch.cuda.synchronize()
        w2.adopt_live_state()
        assert w2.len == w.len and w2.frame == w.frame
        cm.assert_states_equal(cm.snapshot_state(w2, ids2), cm.snapshot_state(w, ids), "adopted")
        assert w2.save() == w.save()
    finally:
        dist.destroy_process_group()
