"""The HIP path against the committed known-answer vectors -- no oracle involved at run time."""
import pytest

import bevy_ggrs_amd as bg
from golden_util import GOLDEN, replay_particles_case, replay_scenario

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, bg.GGRS_WORLD_NT_COPY, bg.GGRS_WORLD_NO_GROUPS, bg.GGRS_WORLD_UNFUSED])
@pytest.mark.parametrize("name", sorted(GOLDEN["particles_synctest"]))
def test_particles_synctest_vectors(name, flags):
    replay_particles_case(lambda cap, depth: bg.World(cap, max_depth=depth, flags=flags), GOLDEN["particles_synctest"][name])


SCEN = [(kind, name) for kind in sorted(GOLDEN["scenarios"]) for name in sorted(GOLDEN["scenarios"][kind])]


@pytest.mark.parametrize("flags", [0, bg.GGRS_WORLD_NO_GROUPS])
@pytest.mark.parametrize("kind,name", SCEN)
def test_scenario_vectors(kind, name, flags):
    replay_scenario(lambda cap, depth: bg.World(cap, max_depth=depth, flags=flags), kind, GOLDEN["scenarios"][kind][name])
