"""RollbackOrdered (src/snapshot/rollback.rs:62-99): slot == insertion-order index on this path.  The reference's own
unit tests (rollback.rs:119-196) re-expressed over `spawn` / `len` / snapshots, for the oracle (CPU) and the HIP world."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
from oracle.binding import FLAT, REFSHAPED, OracleWorld


def _world(make):
    w = make(64, 8)
    c = w.register_component("Tag", 4, 1)
    w.checksum_component(c, [0])
    return w, c


def _scenario(make):
    w, c = _world(make)
    # order_returns_insertion_index (rollback.rs:121-127): ids receive zero-based, insertion-order indices
    assert w.spawn(1, {c: [np.array([10], np.uint32)]}) == 0
    assert w.spawn(1, {c: [np.array([20], np.uint32)]}) == 1
    assert w.spawn(1, {c: [np.array([30], np.uint32)]}) == 2
    # iter_sorted_yields_insertion_order (:130-135): the column read back in slot order IS the insertion order
    assert list(w.download_word(c, 0, 0, 3)) == [10, 20, 30]
    # order_is_stable_after_more_pushes (:138-147)
    cs3 = w.save()
    assert w.spawn(2, {c: [np.array([40, 50], np.uint32)]}) == 3
    assert list(w.download_word(c, 0, 0, 3)) == [10, 20, 30] and w.len == 5
    # clone_is_independent (:187-195) + RollbackOrdered restored by LoadWorld (mod.rs:342): the snapshot's copy has
    # len 3; loading it forgets the later ids, and the next spawn reuses index 3
    w.advance(); w.save()
    w.load(0)
    assert w.len == 3 and w.save() == cs3
    assert w.spawn(1, {c: [np.array([60], np.uint32)]}) == 3
    assert list(w.download_word(c, 0, 0, 4)) == [10, 20, 30, 60]
    # despawned ids stay registered (rollback.rs:60-66 "including despawned entities"): their index is never reused
    w.despawn(1)
    assert w.spawn(1, {c: [np.array([70], np.uint32)]}) == 4 and w.len == 5 and w.active_count() == 4
    return w


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_rollback_ordered(mode):
    w = _scenario(lambda cap, depth: OracleWorld(cap, depth, mode))
    with pytest.raises(Exception):
        w.despawn(99)                                   # order_unregistered_panics (:150-155)


@pytest.mark.gpu
def test_gpu_rollback_ordered():
    w = _scenario(lambda cap, depth: bg.World(cap, max_depth=depth))
    with pytest.raises(bg.GgrsHipError):
        w.despawn(99)
