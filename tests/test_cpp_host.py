"""The C++ host mirror of the reference's plugin interface (include/bevy_ggrs_hip.hpp), driven by
tests/cpp/host_test.cpp -- the reference's own integration tests (tests/synctest.rs,
tests/component_rollback.rs, tests/common/mod.rs) re-expressed against that mirror.

CPU: host logic (SessionBuilder / SyncTestSession / run_ggrs_schedules accumulator /
handle_requests marshalling) on the oracle backend.  GPU: the same program on libggrs_hip.so; its
output (every Checksum(u128) of every SaveGameState, final counters, a fold of translation.y) must
equal the oracle build's byte for byte."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_test.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "_build")


def _build(kind):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, f"host_test_{kind}")
    deps = [SRC, os.path.join(ROOT, "include", "bevy_ggrs_hip.hpp"), os.path.join(ROOT, "include", "ggrs_hip.h")]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", f"-I{ROOT}/include", SRC, "-o", exe]
    if kind == "oracle":
        d = os.path.join(ROOT, "oracle", "_build")
        cmd += ["-DBACKEND_ORACLE", f"-L{d}", "-lggrs_oracle", f"-Wl,-rpath,{d}"]
    else:
        d = os.path.join(ROOT, "bevy_ggrs_amd")
        cmd += [f"-L{d}", "-lggrs_hip", f"-Wl,-rpath,{d}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def _run(exe, n):
    # the oracle's OpenMP loops must not spin up one thread per host core of the GPU box (256)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    out = subprocess.run([exe, str(n)], check=True, capture_output=True, text=True, timeout=600, env=env).stdout
    # librccl prints a banner to stdout when the product build's fan-out test initialises its communicator
    return "".join(l for l in out.splitlines(keepends=True) if not l.startswith(("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path")))


EXPECTED = ["synctest_request_shape", "despawn_and_rollback_does_not_panic", "mismatch_fires_on_non_determinism",
            "confirmed_frame_pruning", "component_rollback_copy", "immutable_component_copy_strategy_rolls_back", "fixed_timestep_accumulator", "ggrs_time_survives_session_restart", "host_seahasher_known_answers",
            "host_ring_known_answers", "resource_inserted_mid_session_rolls_back", "resource_removed_mid_session_rolls_back",
            "resource_without_rollback_fires_mismatch", "resource_checksum_part_is_folded", "box_game_synctest", "particles",
            "particles_pipelined", "speculative_fanout_adopts_matching_branch"]


def test_cpp_host_on_oracle_backend():
    out = _run(_build("oracle"), 3000)
    for name in EXPECTED:
        assert f"ok {name}" in out
    assert out.count("\nchecksum ") == 2 * (8 + 16 * 7)      # cd = 7: frames 0..7 save once, then 7 saves per tick; sync + pipelined runs
    assert out.count("box checksum ") == 8 + 32 * 7       # box_game: cd = 7, 40 updates
    lines = [l for l in out.splitlines() if l.startswith(("checksum", "final"))]
    assert lines[:len(lines) // 2] == lines[len(lines) // 2:], "pipelined run must reproduce the synchronous run"


def test_cpp_host_builds_against_the_product_library():
    """Link check only (no GPU here): the mirror's HipBackend binds every C-ABI symbol it uses."""
    _build("hip")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 70_000])
def test_cpp_host_on_hip_matches_oracle_build(n):
    want = _run(_build("oracle"), n)
    got = _run(_build("hip"), n)
    assert got == want


def test_synctest_sessions_agree():
    """The C++ (include/bevy_ggrs_hip.hpp) and Python (bevy_ggrs_amd/session.py) restatements of ggrs's
    SyncTestSession::advance_frame emit the same request lists -- kinds, frames and (delayed) inputs -- for check
    distances 0/1/3/7 and input delays 0/2.  (ggrs itself is un-vendored: its request order is parity-unpinned by the
    reference; two independent restatements that agree is what can be checked here.)"""
    import bevy_ggrs_amd as bg
    from bevy_ggrs_amd.session import SyncTestSession
    out = _run(_build("oracle"), 500)
    got = {}
    for l in out.splitlines():
        if l.startswith("trace "):
            head, _, body = l.partition(":")
            got[head] = body.split()
    assert len(got) == 4 * 2 * 14
    for cd in (0, 1, 3, 7):
        for delay in (0, 2):
            s = SyncTestSession(2, cd, 8, delay)
            for t in range(14):
                for h in range(2):
                    s.add_local_input(h, (t * 5 + h * 3) & 15)
                want = []
                reqs = s.advance_frame()
                for r in reqs:
                    if isinstance(r, bg.SaveGameState): want.append(f"S{r.frame}")
                    elif isinstance(r, bg.LoadGameState): want.append(f"L{r.frame}")
                    else: want.append("A" + ",".join(str(int(i)) for i in r.inputs))
                s.record_checksums([r.frame * 7 + 1 for r in reqs if isinstance(r, bg.SaveGameState)])
                assert got[f"trace cd={cd} delay={delay} t={t}"] == want, (cd, delay, t)
