"""Every environment knob the library reads (csrc/host_world.hpp `Knobs`; INTEGRATION.md lists them with the measurement that keeps each)
selects among kernels / policies that must all produce the same bits: one small and one HBM-sized bit-exact parity case against the CPU
oracle under each setting -- blocking lists AND pipelined (enqueue / collect) ticks -- so the driver-run suite covers every kernel that
ships: the generated kernel with its rows folded by the host / forwarded to the next launch (fold-forward) / by k_gen_finalize, with and
without row versions and specialised copies, and the per-request kernels (no run-time compiler, no shipped code object).
(GGRS_HIP_ROCTX=1 needs the roctx library of a profiler session and has no case; GGRS_AOT_DIR's positive case is tests/test_gpu_aot.py.)"""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
import common as cm
from oracle.binding import FLAT, OracleWorld

pytestmark = pytest.mark.gpu

KNOBS = [
    {},                                                     # the defaults
    {"GGRS_TICK_JIT": "0"},                                 # no generated kernel: one launch per request
    {"GGRS_NO_HIPRTC": "1", "GGRS_AOT_DIR": "0", "GGRS_JIT_CACHE_DIR": "0"},   # a deployment without libhiprtc.so and without shipped objects: the same fallback, found by itself
    {"GGRS_FOLD_FORWARD_MIN_WGS": "0"},                     # every enqueued group's rows are folded by the next launch (even the 10 k world's 40)
    {"GGRS_FOLD_FORWARD_MIN_WGS": "1000000"},               # never: the host folds one row per workgroup at collect time (rounds 3-4), blocking calls above 1024 workgroups k_gen_finalize
    {"GGRS_STAGE_BYTES": "4096"},                           # a spawn-payload ring of 4 KiB: the lists' spawns wrap it (a full ring waits for the stream)
    {"GGRS_HIP_TRACE": "1", "GGRS_DEBUG_JIT": "1"},         # the diagnostic prints change nothing
    {"GGRS_ROW_VERSIONS": "0"},
    {"GGRS_JIT_SPECIALISE_AFTER": "1", "GGRS_JIT_SPECIALISE_SYNC": "1"},   # every group shape gets its own kernel at once (built on the calling thread)
    {"GGRS_JIT_SPECIALISE_AFTER": "0"},                     # never
    {"GGRS_SPIN_WAIT_US": "0"},                             # no polling of completion tags / events: the runtime's waits, as in rounds 1-3
    {"GGRS_SPIN_WAIT_US": "1"},                             # ... and polls that give up at once (fall back to the runtime's wait mid-flight)
    {"GGRS_JIT_CACHE_DIR": "0"},                            # no code objects on disk
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
    # pipelined ticks (one in flight): rollbacks of 3 frames with a spawn in every second tick -- the enqueue / collect API, i.e. the fold-forward path on
    # the device side; the oracle runs the same lists synchronously
    def tick(F, k):
        reqs = [bg.LoadGameState(F - 3)]
        for i in range(3):
            a = bg.AdvanceFrame((cm.INPUT_SPAWN if (k % 2 == 0 and i == 1) else 0,))
            if a.inputs[0]: a.spawn_vx, a.spawn_vy = fn(F - 3 + i)
            reqs += [a, bg.SaveGameState(F - 2 + i)]
        return reqs + [bg.AdvanceFrame((0,))]
    F = world.frame
    out += world.handle_requests([bg.AdvanceFrame((0,)), bg.SaveGameState(F + 1), bg.AdvanceFrame((0,)), bg.SaveGameState(F + 2), bg.AdvanceFrame((0,)), bg.SaveGameState(F + 3)])
    F = world.frame
    world.set_confirmed(F - 3)
    if hasattr(world, "enqueue_requests"):
        world.enqueue_requests(tick(F, 0))
        for k in range(1, 9):
            world.enqueue_requests(tick(F + k, k)); out += world.collect_checksums()
        out += world.collect_checksums()
    else:
        for k in range(9): out += world.handle_requests(tick(F + k, k))
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
    per_request = env.get("GGRS_TICK_JIT") == "0" or (env.get("GGRS_NO_HIPRTC") == "1" and info["generated_kernel"] != "ok")   # (a module another world of this process built serves a world of the same shape)
    if per_request: assert k.startswith("per-request"), k
    else: assert k.startswith("ggrs_jit_tick"), k
    if not per_request:
        big = n > 300_000
        want = "fold-forward" if (env.get("GGRS_FOLD_FORWARD_MIN_WGS") == "0" or (big and "GGRS_FOLD_FORWARD_MIN_WGS" not in env)) else "the host folds"
        assert info["checksum_fold"].startswith(want), info
        assert info["spawn_system"].startswith("runs inside"), info
        assert int(info["kernarg_bytes"]) < 1000, info                                       # the per-world argument block (the one-size block of rounds 2-4 was 2112 bytes)
    if env.get("GGRS_NO_HIPRTC") == "1": assert info["hiprtc"].startswith("missing") and ("no shipped code object" in info["generated_kernel"] or info.get("generated_kernel_origin", "").startswith("in-process")), info
    if env.get("GGRS_SPIN_WAIT_US") == "0": assert info["blocking_wait"].startswith("hipStreamSynchronize"), info


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


def test_fold_placement_by_world_size():
    """Default policy of the checksum fold by world size (host_world.hpp Knobs::fold_forward_min_wgs): up to 1024 workgroups (262 k slots) one row per workgroup goes
    to the host, beyond the next launch folds the rows (fold-forward) -- at every size, 4 M slots included (the on-chip group fold of round 4 is gone)."""
    for n, want in ((100_000, "the host folds"), (300_000, "fold-forward"), (1_000_000, "fold-forward"), (3_300_000, "fold-forward")):
        w = bg.World(n, max_depth=2)
        ids = cm.build_particles(w)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        info = w.kernel_info()
        w.close()
        assert info["checksum_fold"].startswith(want), (n, info["checksum_fold"])
