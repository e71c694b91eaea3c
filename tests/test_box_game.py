"""box_game (BASELINE config 1): move_cube_system
(examples/box_game/box_game.rs:154-206) as a kernel-backed GgrsSchedule system.

CPU: the C++ oracle (both storage shapes) against the independent numpy twin, and the SyncTest walk of
box_game_synctest.rs (2 players, check distance 7, input delay 2) -- the "CPU reference path" config.
GPU: k_box_move against the oracle, bit for bit, on worlds of many cubes with random inputs so that
every branch (acceleration, friction, speed clamp, plane clamp) is taken."""
import numpy as np
import pytest

import bevy_ggrs_amd as bg
from oracle import oracle_np as onp
from oracle.binding import FLAT, REFSHAPED, OracleWorld

import common as cm

HALF_WIDTH = float((np.float32(5.0) - np.float32(0.2)) * np.float32(0.5))
BOX_PARAMS = (18.0, 3.0, 0.0018, HALF_WIDTH)          # ACCELERATION, MAX_SPEED, FRICTION, half_width (box_game.rs:18-22,202)


def build_box(world, n, num_players, seed=5, spread=False, checksums=()):
    """setup_system (box_game.rs:89-143): cube `handle` on a circle of radius PLANE_SIZE/4; Transform (clone) and
    Velocity (copy) are registered for rollback, Player is a plain component (box_game_synctest.rs:44-45)."""
    T = world.register_component("Transform", 4, 10)
    V = world.register_component("Velocity", 4, 3)
    P = world.register_component("Player", 8, 1, rollback=False)
    world.set_component_default(T, cm.TRANSFORM_DEFAULT)
    for c, words in checksums: world.checksum_component((T, V, P)[c], words)      # (registration ends with the first spawn)
    world.add_system(bg.SYS_BOX_MOVE, comp=(T, V, P), word=(0, 0, 0), fparam=BOX_PARAMS)
    handle = (np.arange(n) % num_players).astype(np.uint64)
    r = np.float32(5.0) / np.float32(4.0)
    rot = (np.arange(n) % num_players).astype(np.float32) / np.float32(num_players) * np.float32(2.0) * np.float32(np.pi)
    tr = np.tile(cm.TRANSFORM_DEFAULT, (n, 1)).astype(np.float32)
    tr[:, 0] = r * np.cos(rot).astype(np.float32)
    tr[:, 1] = np.float32(0.2) / np.float32(2.0)
    tr[:, 2] = r * np.sin(rot).astype(np.float32)
    vel = np.zeros((n, 3), dtype=np.float32)
    if spread:                                          # beyond the example: arbitrary starting states
        rng = np.random.default_rng(seed)
        tr[:, 0:3] = rng.uniform(-2.6, 2.6, (n, 3)).astype(np.float32)
        vel = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    world.spawn(n, {T: [cm.f32bits(tr[:, k]) for k in range(10)], V: [cm.f32bits(vel[:, k]) for k in range(3)], P: [handle]})
    return (T, V, P), tr[:, 0:3].copy(), vel, handle


def box_state(world, ids):
    T, V, P = ids
    n = world.len
    t = np.stack([world.download_word(T, k, 0, n) for k in range(3)], axis=1)
    v = np.stack([world.download_word(V, k, 0, n) for k in range(3)], axis=1)
    return t, v


def input_script(frame, num_players, seed=11):
    return tuple(int(x) for x in np.random.default_rng([seed, frame]).integers(0, 16, num_players))


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_box_move_matches_numpy_twin(mode):
    n, players = 4096, 4
    w = OracleWorld(n, 8, mode)
    ids, t, v, handle = build_box(w, n, players, spread=True)
    for f in range(1, 40):
        inputs = input_script(f, players)
        w.handle_requests([bg.AdvanceFrame(inputs)])
        t, v = onp.box_move(t, v, np.array(inputs, dtype=np.uint8)[handle], int(onp.dt_bits(60, f)))
        got_t, got_v = box_state(w, ids)
        assert np.array_equal(got_t, cm.f32bits(t).reshape(n, 3)), f"translation differs at frame {f}"
        assert np.array_equal(got_v, cm.f32bits(v).reshape(n, 3)), f"velocity differs at frame {f}"
    # the script drove cubes into every branch
    assert (np.abs(t[:, 0]) == np.float32(HALF_WIDTH)).any() and (np.abs(t[:, 2]) == np.float32(HALF_WIDTH)).any()


def synctest_box_game(world, num_players=2, check_distance=7, ticks=40, n=None):
    """box_game_synctest.rs with `--num-players 2 --check-distance 7` (examples/README.md:64), input delay 2."""
    n = n or num_players
    ids, *_ = build_box(world, n, num_players)
    drv = cm.SyncTestDriver(world, check_distance, num_players=num_players, input_delay=2)
    trace = []
    for tick in range(ticks):
        drv.tick(input_script(tick, num_players))
        trace.append(box_state(world, ids))
    return drv.all_checksums, trace


@pytest.mark.parametrize("mode", [FLAT, REFSHAPED])
def test_oracle_box_game_synctest_2_players_cd7(mode):
    cs, trace = synctest_box_game(OracleWorld(16, 8, mode))
    seen = {}
    for f, c in cs:
        assert seen.setdefault(f, c) == c
    assert len(cs) == 8 + (40 - 8) * 7                  # frames 0..7 once, then check_distance saves per tick
    t0, _ = trace[0]
    tN, vN = trace[-1]
    assert not np.array_equal(t0, tN)                   # the cubes moved
    speed = np.sqrt((vN.view(np.float32).astype(np.float64) ** 2).sum(axis=1))
    assert (speed <= 3.0 + 1e-5).all()


def test_oracle_box_game_shapes_agree():
    a = synctest_box_game(OracleWorld(300, 8, FLAT), num_players=4, check_distance=3, ticks=25, n=300)
    b = synctest_box_game(OracleWorld(300, 8, REFSHAPED), num_players=4, check_distance=3, ticks=25, n=300)
    assert a[0] == b[0]
    for (ta, va), (tb, vb) in zip(a[1], b[1]):
        assert np.array_equal(ta, tb) and np.array_equal(va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("n,players,cd,ticks", [(2, 2, 7, 40), (4, 4, 2, 20), (5000, 4, 3, 30), (70_000, 16, 7, 24)])
def test_gpu_box_game_synctest_matches_oracle(n, players, cd, ticks):
    got = synctest_box_game(bg.World(n + 8, max_depth=8), players, cd, ticks, n)
    want = synctest_box_game(OracleWorld(n + 8, 8), players, cd, ticks, n)
    assert got[0] == want[0]
    for tick, ((ta, va), (tb, vb)) in enumerate(zip(got[1], want[1])):
        assert np.array_equal(ta, tb), f"translation differs at tick {tick}"
        assert np.array_equal(va, vb), f"velocity differs at tick {tick}"


@pytest.mark.gpu
def test_gpu_box_move_every_branch_matches_oracle():
    n, players = 20_000, 4
    res = []
    for w in (bg.World(n, max_depth=8), OracleWorld(n, 8)):
        ids, *_ = build_box(w, n, players, spread=True)
        out = []
        for f in range(1, 30):
            w.handle_requests([bg.AdvanceFrame(input_script(f, players))])
            out.append(box_state(w, ids))
        res.append(out)
    for f, ((ta, va), (tb, vb)) in enumerate(zip(*res)):
        assert np.array_equal(ta, tb) and np.array_equal(va, vb), f"frame {f + 1}"
