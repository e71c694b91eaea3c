"""Entity references through rollback (tests/hierarchy.rs, src/snapshot/childof_snapshot.rs, rollback_entity_map.rs).

In the reference LoadWorld re-creates despawned entities with NEW Bevy ids, so every stored `Entity` (ChildOf, anything
implementing MapEntities) has to be remapped through `RollbackEntityMap`.  On this path an entity reference is a SLOT
(== RollbackOrdered index): stable, never reused, restored by LoadWorld -- the entity map is the identity and a link is
just a registered 8-byte word.  These tests re-express tests/hierarchy.rs with that representation: the link column is
checksummed, so a link that did not survive a rollback would raise SyncTestMismatch."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, REFSHAPED, OracleWorld

NO_PARENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def build(world, delete_child_at=None):
    """ParentEntity / ChildEntity / GrandchildEntity markers, `ChildOf(parent slot)`, and a `Life` countdown that stands
    in for delete_child_system (hierarchy.rs:37-46): the child despawns when its Life reaches zero."""
    marker = world.register_component("Marker", 4, 1)         # 1 = parent, 2 = child, 3 = grandchild
    child_of = world.register_component("ChildOf", 8, 1)
    life = world.register_component("Life", 4, 1)
    for c in (marker, child_of, life):
        world.checksum_component(c, [0])
    world.add_system(bg.SYS_SAT_SUB_DESPAWN, comp=(life,), word=(0,), iparam=(1, bg.DESPAWN_IMMEDIATE))
    parent = world.spawn(1, {marker: [np.array([1], np.uint32)]})                      # no ChildOf, no Life: lives forever
    child = world.spawn(1, {marker: [np.array([2], np.uint32)], child_of: [np.array([parent], np.uint64)],
                            **({life: [np.array([delete_child_at], np.uint32)]} if delete_child_at else {})})
    grandchild = world.spawn(1, {marker: [np.array([3], np.uint32)], child_of: [np.array([child], np.uint64)]})
    return (marker, child_of, life), (parent, child, grandchild)


def links(world, ids):
    marker, child_of, _ = ids
    n = world.len
    alive = world.alive_mask(n)
    has = world.present_mask(child_of, n) & alive
    return alive, np.where(has, world.download_word(child_of, 0, 0, n), NO_PARENT), world.download_word(marker, 0, 0, n)


def run(world, updates, cd=2, delete_child_at=None):
    ids, slots = build(world, delete_child_at)
    drv = cm.SyncTestDriver(world, cd)
    for _ in range(updates):
        drv.tick((0,))                                         # SyncTestDriver raises on a checksum mismatch
    return drv.all_checksums, links(world, ids), slots


def check_recursive(res):
    # recursive_hierarchy_is_preserved_through_rollback (hierarchy.rs:51-112): 20 updates at check distance 2
    _, (alive, parent_of, marker), (p, c, g) = res
    assert alive[[p, c, g]].all()
    assert parent_of[p] == NO_PARENT and parent_of[c] == p and parent_of[g] == c
    assert list(marker[[p, c, g]]) == [1, 2, 3]


def check_deleted(res):
    # hierarchy (hierarchy.rs:114-181): the child is deleted mid-session, rollbacks cross the deletion frame
    _, (alive, parent_of, _), (p, c, g) = res
    assert alive[p] and not alive[c]
    assert alive[g] and parent_of[g] == c                      # a link to a despawned slot stays what it was


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_hierarchy_links(mode):
    check_recursive(run(OracleWorld(16, 8, mode), 20))
    check_deleted(run(OracleWorld(16, 8, mode), 8, delete_child_at=3))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, bg.GGRS_WORLD_NO_GROUPS])
def test_gpu_hierarchy_links_match_oracle(flags):
    for updates, delete_at, check in ((20, None, check_recursive), (8, 3, check_deleted)):
        got = run(bg.World(16, max_depth=8, flags=flags), updates, delete_child_at=delete_at)
        want = run(OracleWorld(16, 8), updates, delete_child_at=delete_at)
        check(got)
        assert got[0] == want[0]
        for a, b in zip(got[1], want[1]):
            assert np.array_equal(a, b)
