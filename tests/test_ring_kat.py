"""The 11 GgrsSnapshots unit tests of /root/reference/src/snapshot/mod.rs:365-512, mirrored 1:1
against the oracle's ring.  These are the reference's own known-answer tests for the snapshot
ring; the HIP library's ring is checked against the same list in test_gpu_ring.py."""
import pytest

from oracle.binding import OracleRing

I32_MAX, I32_MIN = 2**31 - 1, -2**31


def snap_with_depth(d):
    return OracleRing(d)


def test_push_evicts_oldest_when_depth_exceeded():      # mod.rs:369-380
    s = snap_with_depth(3)
    for i in range(5): s.push(i, i)
    assert s.peek(0) is None and s.peek(1) is None
    assert (s.peek(2), s.peek(3), s.peek(4)) == (2, 3, 4)


def test_push_older_frame_discards_newer():             # mod.rs:384-394
    s = snap_with_depth(8)
    s.push(5, 50); s.push(6, 60); s.push(7, 70)
    s.push(5, 99)
    assert s.peek(5) == 99 and s.peek(6) is None and s.peek(7) is None


def test_push_same_frame_replaces():                    # mod.rs:398-403
    s = snap_with_depth(8)
    s.push(3, 10); s.push(3, 20)
    assert s.peek(3) == 20


def test_confirm_prunes_older_frames():                 # mod.rs:409-422
    s = snap_with_depth(8)
    for i in range(6): s.push(i, i)
    s.confirm(3)
    assert s.peek(0) is None and s.peek(1) is None and s.peek(2) is None
    assert (s.peek(3), s.peek(4), s.peek(5)) == (3, 4, 5)


def test_confirm_beyond_all_frames_empties_storage():   # mod.rs:426-435
    s = snap_with_depth(8)
    for i in range(4): s.push(i, i)
    s.confirm(100)
    assert all(s.peek(i) is None for i in range(4))


def test_confirm_on_empty_does_not_panic():             # mod.rs:439-442
    snap_with_depth(8).confirm(5)


def test_rollback_to_existing_frame():                  # mod.rs:448-455
    s = snap_with_depth(8)
    for i in range(5): s.push(i, i * 10)
    s.rollback(2)
    assert s.get() == 20


def test_rollback_discards_newer_frames():              # mod.rs:459-468
    s = snap_with_depth(8)
    for i in range(5): s.push(i, i)
    s.rollback(2)
    assert s.peek(3) is None and s.peek(4) is None and s.peek(2) == 2


def test_rollback_missing_frame_panics():               # mod.rs:472-477
    s = snap_with_depth(8)
    s.push(0, 0)
    with pytest.raises(RuntimeError, match="Could not rollback to 99"):
        s.rollback(99)


def test_push_wraps_i32_max_to_min_retains_history():   # mod.rs:485-497
    s = snap_with_depth(8)
    s.push(I32_MAX - 2, 1); s.push(I32_MAX - 1, 2); s.push(I32_MAX, 3)
    s.push(I32_MIN, 4)
    assert (s.peek(I32_MAX - 2), s.peek(I32_MAX - 1), s.peek(I32_MAX), s.peek(I32_MIN)) == (1, 2, 3, 4)


def test_push_max_after_min_evicts_min_as_future():     # mod.rs:502-512
    s = snap_with_depth(8)
    s.push(I32_MIN, 1)
    s.push(I32_MAX, 2)
    assert s.peek(I32_MIN) is None
    assert s.peek(I32_MAX) == 2
