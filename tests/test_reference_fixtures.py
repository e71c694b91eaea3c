"""Parity against the REAL bevy_ggrs.  tests/golden/reference_checksums.json is produced by the Rust
fixture run (rust/fixtures: real bevy_ggrs 0.22 + ggrs under the reference's own SyncTest harness, fed this repo's synthetic
inputs).  No Rust toolchain exists in the builder's image, so the file is absent there and these tests SKIP; the day it
exists they are what turns "parity unpinned by the reference" into a pinned oracle: every Checksum(u128) the reference saved
for a frame must equal the CPU oracle's -- and, under -m gpu, the HIP path's -- for that frame."""
import json
import os

import pytest

import bevy_ggrs_amd as bg
import common as cm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_checksums.json")
CASES = {"config2_particles_10k": 10_000, "config3_particles_1m": 1_000_000}


def _reference(case):
    if not os.path.exists(GOLDEN):
        pytest.skip("tests/golden/reference_checksums.json is absent: run rust/fixtures (cargo test --release) on a machine with a Rust toolchain")
    ref = json.load(open(GOLDEN))[case]
    first = {}
    for f, c in ref["saves"]:
        assert first.setdefault(f, int(c, 16)) == int(c, 16), "the reference itself resimulated a frame to a different checksum"
    return ref, first


def _ours(world, n, ticks):
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ids = cm.build_particles(world)
    cm.spawn_particles(world, ids, n, vel, ttl)
    drv = cm.SyncTestDriver(world, 8, max_prediction=9)
    for _ in range(ticks): drv.tick((0,))
    return dict(drv.all_checksums)          # resimulated frames agree with their first save (SyncTest asserts it)


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_equals_the_reference(case):
    from oracle.binding import FLAT, OracleWorld
    ref, first = _reference(case)
    got = _ours(OracleWorld(CASES[case], 9, FLAT), CASES[case], ref["ticks"])
    assert set(first) <= set(got) | {max(first)}, "the reference saved frames we never did"
    for f, c in first.items():
        if f in got: assert got[f] == c, f"frame {f}: ours {got[f]:#x} reference {c:#x}"


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_equals_the_reference(case):
    ref, first = _reference(case)
    got = _ours(bg.World(CASES[case], max_depth=9), CASES[case], ref["ticks"])
    for f, c in first.items():
        if f in got: assert got[f] == c, f"frame {f}: HIP {got[f]:#x} reference {c:#x}"
