"""Every environment knob the library reads (csrc/host_world.hpp `Knobs`) selects among kernels / policies that must all
produce the same bits: one small and one HBM-sized bit-exact parity case against the CPU oracle under each setting, so the
driver-run suite covers every kernel that ships -- the generated kernel in its per-tile and persistent forms with the fold
on the host / in k_gen_finalize / in the launch (tick_fold) / per group of 64 workgroups, with and without depth-parallel
roles, dead-snapshot elimination and row versions, the per-request kernels (no run-time compiler), on paged and contiguous arenas.  (Two knobs have no case: GGRS_ARENA_PARK=0
re-enables the runtime hazard of profiles/r03fc and exists for that experiment only; GGRS_HIP_ROCTX=1 needs the roctx library of a
profiler session.)"""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu

KNOBS = [
    {},                                                     # the defaults
    {"GGRS_TICK_JIT": "0"},                                 # no generated kernel (a deployment without libhiprtc.so): one launch per request
    {"GGRS_JIT_PERSIST_MIN_SLOTS": "1"},                    # generated kernel, persistent form + in-launch fold even for small worlds
    {"GGRS_HOST_FOLD_MAX_WGS": "0"},                        # every group folded on the device (k_gen_finalize)
    {"GGRS_HOST_FOLD_MAX_WGS": "256"},                      # only small groups folded by the host (the round-2 default)
    {"GGRS_GROUP_FOLD_MIN_WGS": "0"},                       # no group fold: one row per workgroup leaves the kernel at every size
    {"GGRS_GROUP_FOLD_MIN_WGS": "8", "GGRS_JIT_DP": "0"},   # group fold even for the 10 k world (one group of 56 workgroups incl. a padding one)
    {"GGRS_GROUP_FOLD_MIN_WGS": "8", "GGRS_JIT_DP": "0", "GGRS_HOST_FOLD_MAX_WGS": "0"},   # ... with the groups' rows staying on the device (k_gen_finalize over rows / 64)
    {"GGRS_JIT_NT_LOADS": "1"},                            # the source block is always loaded non-temporally (default: only when it is not expected in the caches)
    {"GGRS_JIT_NT_LOADS": "0"},
    {"GGRS_JIT_FUSE_SPAWN": "0"},                          # a firing spawn system ends the request group (rounds 1-3: k_spawn_particles + mask edits as their own launches)
    {"GGRS_DEAD_GROUPS": "0"},
    {"GGRS_JIT_PERSIST_MIN_SLOTS": "1", "GGRS_JIT_PERSIST_OVERSUB": "4", "GGRS_JIT_PERSIST_TPB": "512"},   # persistent form, another grid shape
    {"GGRS_JIT_PERSIST_MIN_SLOTS": "1", "GGRS_JIT_PERSIST_OVERSUB": "64", "GGRS_JIT_PERSIST_TPB": "256"},  # ... and one whose grid must be clamped to tick_fold's row buffer
    {"GGRS_JIT_DP_MAX_SLOTS": "400000"},                    # depth-parallel roles far above their default range
    {"GGRS_HIP_TRACE": "1", "GGRS_DEBUG_JIT": "1", "GGRS_DEBUG_ARENA": "1"},   # the diagnostic prints change nothing
    {"GGRS_JIT_DP": "0"},
    {"GGRS_JIT_DP": "3"},
    {"GGRS_ROW_VERSIONS": "0"},
    {"GGRS_JIT_SPECIALISE_AFTER": "1", "GGRS_JIT_SPECIALISE_SYNC": "1"},   # every HBM-sized group shape gets its own kernel at once (built on the calling thread)
    {"GGRS_JIT_SPECIALISE_AFTER": "0"},                     # never
    {"GGRS_JIT_SPECIALISE_AFTER": "1", "GGRS_JIT_SPECIALISE_SYNC": "1", "GGRS_JIT_SPEC_SHAPES": "1"},   # one place in the shape table: every new shape unloads the previous kernel
    {"GGRS_JIT_LANE_FOLD": "1"},                            # checksum fold through per-lane LDS rows even for small worlds
    {"GGRS_JIT_LANE_FOLD": "0"},                            # ... and the per-Save DPP ladder even for big ones
    {"GGRS_EVENT_ON_KERNEL": "0"},                          # an enqueued list ends with a marker packet again
    {"GGRS_HOST_FOLD_MAX_WGS": "0", "GGRS_SPIN_WAIT_US": "0"},   # every fold by k_gen_finalize (blocking calls then poll its completion tags, default 200 us) with the poll off:
                                                                 # hipStreamSynchronize per blocking call, as in rounds 1-3
    {"GGRS_HOST_FOLD_MAX_WGS": "0", "GGRS_SPIN_WAIT_US": "1"},   # ... and a poll that gives up at once (falls back to the stream wait mid-flight)
    {"GGRS_PRESENCE_VERSIONS": "0"},                        # presence masks stored with every Save
    {"GGRS_JIT_CACHE_FIRST_SAVE": "0"},                     # every Save of an HBM-sized rollback group streams past the caches
    {"GGRS_ARENA_CONTIG": "0"},
    {"GGRS_ARENA_CONTIG": "1"},                             # every world on a physically contiguous arena (parked when its world closes, reused by the next)
    {"GGRS_ARENA_CONTIG": "2", "GGRS_ARENA_FLUSH": "7"},    # particles worlds contiguous + the L2 write-back / invalidate kernels of the r03fc experiment
    {"GGRS_DEBUG_POISON": "1"},
]


def _case(world, n):
    """SyncTest ticks with spawns and Ttl despawns, then a branch list (dead groups, batches), then more ticks."""
    D = 4
    vel, ttl = cm.synthetic_particles(n, ttl="despawn")
    ids = cm.build_particles(world, with_spawn=True, ttl_init=6)
    cm.spawn_particles(world, ids, n, vel, ttl)
    fn = cm.frame_spawn_fn(70)
    drv = cm.SyncTestDriver(world, D, max_prediction=D + 1)
    for t in range(7):
        drv.tick((cm.INPUT_SPAWN if t % 3 == 1 else 0,), spawn_fn=fn)
    out = list(drv.all_checksums)
    C = world.frame - 1                                       # newest snapshot
    world.set_synctest_check_distance(-1)                     # from here on both backends get the same explicit ConfirmedFrameCount
    world.set_confirmed(max(0, C - D))
    reqs = []
    for b in range(4):
        reqs += [bg.LoadGameState(C)]
        for i in range(3):
            reqs += [bg.AdvanceFrame((0,)), bg.SaveGameState(C + 1 + i)]
        reqs.append(bg.AdvanceFrame((0,)))
    out += world.handle_requests(reqs)
    out += world.handle_requests([bg.SaveGameState(world.frame)])
    return out, cm.snapshot_state(world, ids)


_ORACLE = {}


@pytest.mark.parametrize("n", [10_000, 700_000])
@pytest.mark.parametrize("env", KNOBS, ids=lambda e: ",".join(f"{k[5:]}={v}" for k, v in e.items()) or "defaults")
def test_every_knob_keeps_the_bits(env, n, monkeypatch):
    for k, v in env.items(): monkeypatch.setenv(k, v)
    if n not in _ORACLE:
        o = OracleWorld(n + 4000, 6, FLAT)
        _ORACLE[n] = _case(o, n)
        o.close()
    w = bg.World(n + 4000, max_depth=6)
    got = _case(w, n)
    info = w.kernel_info()
    w.close()
    assert got[0] == _ORACLE[n][0], (env, info)
    cm.assert_states_equal(got[1], _ORACLE[n][1], f"{env} n={n}")
    # the knob actually selected what it names
    k = info["request_group_kernel"]
    if env.get("GGRS_TICK_JIT") == "0": assert k.startswith("per-request"), k
    if env.get("GGRS_JIT_PERSIST_MIN_SLOTS") == "1": assert "persistent" in k, k
    if not env: assert k.startswith("ggrs_jit_tick") and "persistent" not in k, k        # the default at every size (host_world.hpp: measured)
    if not env: assert info["checksum_fold"].startswith("the host folds"), info           # (the group fold's default range starts at 12288 workgroups: test_group_fold_default_range)
    if env.get("GGRS_TICK_JIT") != "0" and "GGRS_JIT_PERSIST_MIN_SLOTS" not in env:
        assert info["spawn_system"].startswith("ends the request group" if env.get("GGRS_JIT_FUSE_SPAWN") == "0" else "runs inside"), info
    if env.get("GGRS_GROUP_FOLD_MIN_WGS") == "0": assert "group fold" not in info["checksum_fold"], info
    if env.get("GGRS_GROUP_FOLD_MIN_WGS") == "8": assert info["checksum_fold"].startswith("group fold") and (("k_gen_finalize" in info["checksum_fold"]) == ("GGRS_HOST_FOLD_MAX_WGS" in env)), info
    if env == {"GGRS_HOST_FOLD_MAX_WGS": "0"}: assert " 0 calls so far" not in info["blocking_wait"] and info["blocking_wait"].startswith("polls"), info
    if env.get("GGRS_SPIN_WAIT_US") == "0": assert info["blocking_wait"].startswith("hipStreamSynchronize"), info
    if env.get("GGRS_ARENA_CONTIG") in ("1", "2"): assert info["arena"].startswith("contiguous"), info
    if env.get("GGRS_ARENA_CONTIG") == "0": assert info["arena"].startswith("paged"), info
    if env == {"GGRS_ROW_VERSIONS": "0"}: assert k.startswith("ggrs_jit_tick"), k


def test_missing_runtime_compiler_is_a_queryable_state(monkeypatch):
    """GGRS_TICK_JIT=0 stands in for a deployment without libhiprtc.so: every world falls back to one launch per request (the particles
    schedule with its fused step kernel), and ggrs_hip_world_kernel_info says so."""
    monkeypatch.setenv("GGRS_TICK_JIT", "0")
    res = []
    for w in (bg.World(3000, max_depth=9), OracleWorld(3000, 9, FLAT)):
        H = w.register_component("Health", 4, 1)
        w.checksum_component(H, [0])
        w.add_system(bg.SYS_ADD_U32, comp=(H,), word=(0,), iparam=(3,))
        w.spawn(2500, {H: [np.arange(2500, dtype=np.uint32)]})
        drv = cm.SyncTestDriver(w, 3)
        for _ in range(8): drv.tick((0,))
        res.append(drv.all_checksums)
        if isinstance(w, bg.World):
            info = w.kernel_info()
            assert info["generated_kernel"].startswith("disabled") and info["request_group_kernel"].startswith("per-request"), info
            assert info["hiprtc"].startswith(("loaded", "missing")), info
    assert res[0] == res[1]


def test_group_fold_default_range():
    """Default policy of the checksum fold by world size (host_world.hpp Knobs::group_fold_min_wgs, profiles/r04b): one row per workgroup to the
    host up to 12288 workgroups, the on-chip group fold beyond (4 M slots: 245 rows of 15625 leave the kernel)."""
    for n, want in ((1_000_000, "the host folds"), (3_300_000, "group fold on the chip")):
        w = bg.World(n, max_depth=2)
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        info = w.kernel_info()
        w.close()
        assert info["checksum_fold"].startswith(want), (n, info["checksum_fold"])
