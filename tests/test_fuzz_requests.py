"""Differential fuzzing of request lists (tests/fuzz_util.py).

CPU tier: the oracle's two storage modes against each other (FLAT columns vs the reference-shaped per-entity snapshots,
oracle/ggrs_oracle.cpp) -- this pins the GENERATOR (every list it emits is accepted, its ring model agrees with the backends
about the current frame) and the oracle's internal consistency.  GPU tier: the HIP library against the oracle on the same
seeds, small worlds around the wave / workgroup / layout-tile boundaries and two HBM-sized ones."""
import pytest

import bevy_ggrs_amd as bg
import fuzz_util
from oracle.binding import FLAT, REFSHAPED, OracleWorld


@pytest.mark.parametrize("seed", range(40))
def test_oracle_modes_agree_on_random_request_lists(seed):
    fuzz_util.run(seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: OracleWorld(sc.capacity, 8, REFSHAPED), n_lists=24)


@pytest.mark.parametrize("seed", range(40))
def test_oracle_modes_agree_on_random_request_lists_generic_worlds(seed):
    fuzz_util.run(seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: OracleWorld(sc.capacity, 8, REFSHAPED), n_lists=24, generic=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(120))
def test_hip_matches_the_oracle_on_random_request_lists(seed):
    fuzz_util.run(seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: bg.World(sc.capacity, max_depth=8), n_lists=30)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_hip_matches_the_oracle_on_random_request_lists_hbm_sized(seed):
    fuzz_util.run(1000 + seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: bg.World(sc.capacity, max_depth=8), n_lists=10, big=True, state_every=10)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(160))
def test_hip_matches_the_oracle_on_random_request_lists_generic_worlds(seed):
    fuzz_util.run(seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: bg.World(sc.capacity, max_depth=8), n_lists=30, generic=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_hip_matches_the_oracle_on_random_request_lists_generic_worlds_hbm_sized(seed):
    fuzz_util.run(2000 + seed, lambda sc: OracleWorld(sc.capacity, 8, FLAT), lambda sc: bg.World(sc.capacity, max_depth=8), n_lists=10, big=True, state_every=10, generic=True)
