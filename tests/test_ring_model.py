"""Random operation sequences on the snapshot ring: the C++ oracle's ring against a direct Python transcription of
GgrsSnapshots (src/snapshot/mod.rs:147-243, two VecDeques, newest at the front) -- including frames around the i32
wrap-around, where `push` decides "newer" by `abs_diff > u32::MAX / 2`.  Complements the 11 fixed known-answer tests
(tests/test_ring_kat.py): those pin named cases, this pins everything in between."""
from collections import deque

from hypothesis import given, settings, strategies as st

from oracle.binding import OracleRing

I32_MAX, I32_MIN = 2**31 - 1, -2**31


class ModelRing:
    """Line-by-line model of GgrsSnapshots<For, As> (mod.rs:97-243)."""

    def __init__(self, depth):
        self.snapshots, self.frames, self.depth = deque(), deque(), depth

    def push(self, frame, snapshot):                      # mod.rs:147-181
        while self.frames:
            current = self.frames[0]
            wrapped = abs(current - frame) > (2**32 - 1) // 2
            if (current >= frame and not wrapped) or (frame >= current and wrapped):
                self.snapshots.popleft(); self.frames.popleft()
            else:
                break
        self.snapshots.appendleft(snapshot); self.frames.appendleft(frame)
        while len(self.snapshots) > self.depth:
            self.snapshots.pop(); self.frames.pop()

    def confirm(self, confirmed_frame):                   # mod.rs:185-202
        while self.frames and self.frames[-1] < confirmed_frame:
            self.snapshots.pop(); self.frames.pop()

    def rollback(self, frame):                            # mod.rs:210-226
        while True:
            if not self.frames:
                raise RuntimeError("no snapshot")
            if self.frames[0] != frame:
                self.snapshots.popleft(); self.frames.popleft()
            else:
                return

    def peek(self, frame):                                # mod.rs:236-243
        for f, s in zip(self.frames, self.snapshots):
            if f == frame:
                return s
        return None


frames_near = lambda base: st.integers(min_value=max(I32_MIN, base - 12), max_value=min(I32_MAX, base + 12))
op = lambda base: st.one_of(
    st.tuples(st.just("push"), frames_near(base), st.integers(0, 2**32 - 1)),
    st.tuples(st.just("confirm"), frames_near(base), st.just(0)),
    st.tuples(st.just("rollback"), frames_near(base), st.just(0)),
)


def _run(depth, ops, probe_frames):
    ring, model = OracleRing(depth), ModelRing(depth)
    for kind, frame, value in ops:
        if kind == "push":
            ring.push(frame, value); model.push(frame, value)
        elif kind == "confirm":
            ring.confirm(frame); model.confirm(frame)
        else:
            a = b = None
            try: ring.rollback(frame)
            except RuntimeError as e: a = e
            try: model.rollback(frame)
            except RuntimeError as e: b = e
            assert (a is None) == (b is None), (kind, frame)
        assert len(ring) == len(model.frames)
        for f in probe_frames:
            assert ring.peek(f) == model.peek(f), (kind, frame, f)
        if model.frames:
            assert ring.get() == model.snapshots[0]


@settings(max_examples=300, deadline=None)
@given(depth=st.integers(1, 9), ops=st.lists(op(100), min_size=1, max_size=40))
def test_ring_matches_model_on_ordinary_frames(depth, ops):
    _run(depth, ops, range(86, 115))


@settings(max_examples=300, deadline=None)
@given(depth=st.integers(1, 9), ops=st.lists(st.one_of(op(I32_MAX - 3), op(I32_MIN + 3)), min_size=1, max_size=40))
def test_ring_matches_model_around_the_i32_wrap(depth, ops):
    _run(depth, ops, list(range(I32_MAX - 15, I32_MAX + 1)) + list(range(I32_MIN, I32_MIN + 16)))
