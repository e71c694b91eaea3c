"""bench.py's N > 1 parity gate (fanout_parity) on the CPU: the gathered-table layout SpeculativeFanout keeps (`raw`), the
fast-forward to the first timed step and the serial walk of every branch -- driven here by a fan-out over an ORACLE world
(world size 1, no collectives), so the checker's plumbing is pinned without a GPU.  The GPU leg runs bench.py itself
(tests/test_gpu_zfanout.py::test_bench_two_ranks_line_carries_parity_cpu_baseline_and_roofline)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common as cm  # noqa: E402
from bevy_ggrs_amd.fanout import SpeculativeFanout, default_branch_input  # noqa: E402
from oracle.binding import FLAT, OracleWorld  # noqa: E402


class _OneRank:
    def get_rank(self): return 0
    def get_world_size(self): return 1


class _NoExchange:
    def broadcast(self, dist, src): pass
    def all_gather_u64(self, dist, values): return np.ascontiguousarray(values).reshape(1, -1)


def test_parity_gate_accepts_the_serial_walk_and_names_a_flipped_bit():
    import bench
    n, D, bpr, warm, P = 3000, 4, 3, 5, 2
    w = OracleWorld(n, D + 1, FLAT)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    fan = SpeculativeFanout(w, _OneRank(), D, _NoExchange(), branches_per_rank=bpr)
    fan.sync_confirmed(0)
    for _ in range(warm):
        fan.step(want_result=False)
    c_timed = fan.confirmed
    assert c_timed == warm
    fan.raw, fan.raw_keep = [], P
    for _ in range(P + 2):
        fan.step(want_result=False)
    assert len(fan.raw) == P and [c for c, _ in fan.raw] == [c_timed, c_timed + 1]
    par = bench.fanout_parity(n, D, c_timed, fan.raw, 1, bpr, default_branch_input, lambda f: 0, threads=2)
    assert par["equal"] is True and par["checked_steps"] == P and par["checked_branches"] == bpr and par["checked_saves"] == P * bpr * D, par
    # one flipped bit in one branch's one Save is found and named
    bad = [(c, t.copy()) for c, t in fan.raw]
    bad[1][1][0, 1 * D + 2, 0] ^= np.uint64(1)
    par = bench.fanout_parity(n, D, c_timed, bad, 1, bpr, default_branch_input, lambda f: 0, threads=2)
    assert par["equal"] is False and par["first_mismatch"]["step"] == 1 and par["first_mismatch"]["branch"] == 1 and par["first_mismatch"]["save"] == 2, par
    # a table that starts at the wrong confirmed frame is refused, not compared
    par = bench.fanout_parity(n, D, c_timed + 1, fan.raw, 1, bpr, default_branch_input, lambda f: 0, threads=2)
    assert par["equal"] is False and "confirmed frame" in par["first_mismatch"]["why"], par


def test_shared_prefix_and_per_branch_prefix_gather_the_same_table():
    """SpeculativeFanout(share_prefix=True) computes [Load(C), Advance(confirmed), Save(C+1)] once per step instead of once per branch; the
    canonical table (every branch's D checksums, C+1 first) and the confirmed state must not change -- with spawning inputs too."""
    n, D, bpr, steps = 1500, 4, 3, 5
    out = []
    for share in (True, False):
        w = OracleWorld(n + 100 * (steps + D + 2) * 2, D + 1, FLAT)
        ids = cm.build_particles(w, with_spawn=True, ttl_init=25)
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        cm.spawn_particles(w, ids, n, vel, ttl)
        fan = SpeculativeFanout(w, _OneRank(), D, _NoExchange(), branches_per_rank=bpr, share_prefix=share,
                                branch_input=lambda b, f: cm.INPUT_SPAWN if b % 2 == 0 else 0,
                                confirmed_input=lambda f: cm.INPUT_SPAWN if f % 2 == 1 else 0, spawn_fn=cm.frame_spawn_fn(50))
        assert fan.saves_per_step == (1 + bpr * (D - 1) if share else bpr * D)
        fan.raw_keep = steps
        res = [fan.step() for _ in range(steps)]
        fan.settle()
        out.append((res, [(c, t.tolist()) for c, t in fan.raw], cm.snapshot_state(w, ids)))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    cm.assert_states_equal(out[0][2], out[1][2], "share_prefix")


def test_parity_gate_with_the_spawn_system():
    """bench.py --config 5 --spawn: branches whose predicted input byte carries INPUT_SPAWN (branch ids 16..31, ...) spawn particles every
    frame and diverge; the gate's oracle walk hands the same per-frame spawn payloads to the same branches."""
    import bench
    n, D, bpr, rate = 600, 3, 20, 100
    w = OracleWorld(n + 2 * rate * (D + 2), D + 1, FLAT)
    ids = cm.build_particles(w, with_spawn=True)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    fan = SpeculativeFanout(w, _OneRank(), D, _NoExchange(), branches_per_rank=bpr, spawn_fn=cm.frame_spawn_fn(rate))
    fan.sync_confirmed(0)
    for _ in range(3): fan.step(want_result=False)
    c_timed = fan.confirmed
    fan.raw, fan.raw_keep = [], 2
    res = [fan.step() for _ in range(2)]
    cs = res[-1]["branch_checksums"]
    assert cs[0] == cs[15] and cs[16] == cs[19] and cs[0][0] == cs[16][0] and cs[0][1:] != cs[16][1:]     # spawning branches diverge after the confirmed frame
    par = bench.fanout_parity(n, D, c_timed, fan.raw, 1, bpr, default_branch_input, lambda f: 0, threads=2, spawn_rate=rate)
    assert par["equal"] is True and par["checked_saves"] == 2 * bpr * D, par
    assert bench.fanout_parity(n, D, c_timed, fan.raw, 1, bpr, default_branch_input, lambda f: 0, threads=2, spawn_rate=0)["equal"] is False


def test_alu_view_prices_a_launch_against_the_measured_ceiling():
    """bench.py's informational `roofline.alu`: 1 M entities x 8 Saves x 2 components x 6 diffuse in 49.2 us against profiles/alu_ceiling.json."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    v = b.alu_view(6.0 * 2 * 1_000_000 * 8, 49.2e-6)
    peak = json.load(open(os.path.join(root, "profiles", "alu_ceiling.json")))["diffuse_G_per_s"]
    assert abs(v["achieved"] - 96e6 / 49.2e-6 / 1e9) < 1e-6 and v["peak"] == peak and abs(v["frac"] - v["achieved"] / peak) < 1e-12
    assert 0.5 < v["frac"] < 0.8
    assert b.alu_view(1.0, 0.0)["achieved"] == 0.0
    assert abs(v["executed_over_algorithmic"] - 65 / 96) < 1e-4 and abs(v["frac_executed"] - v["frac"] * 65 / 96) < 1e-6          # 4 of the 6 per component-save + the hoisted order hash


def test_latency_floor_reports_a_lower_bound_against_the_timed_tick_and_keeps_the_same_pass_check():
    """Configs 2 / 4: the timing events slow the loop they ride on, so the same-pass fraction (consistent by construction) prices the instrumented loop; `frac` is the
    bound against the UN-instrumented tick -- (fastest kernel + one boundary) / tick -- and a bound above 1 or a same-pass floor above its tick fails the line."""
    import bench
    lf = bench.latency_floor(6.7, 1.0, 7.77, same_pass=(6.7, 19.39), kernel_min_us=4.52)              # profiles/r06z, config 2 through the C loop
    assert lf["frac_same_pass"] == [round(8.15 / 19.39, 3), round(8.6 / 19.39, 3)] and lf["consistent"]
    assert lf["frac"] == lf["frac_timed_tick_lower_bound"] == round((4.52 + 1.45) / 7.77, 3) and "lower bound" in lf["frac_basis"]
    assert lf["frac_mixed_passes"][0] > 1.0                                                            # the instrumented pass's MEAN kernel does not fit the timed tick: informational
    assert abs(lf["instrumentation_us_per_tick"] - (19.39 - 7.77)) < 0.01
    bad = bench.latency_floor(6.7, 1.0, 5.0, same_pass=(6.7, 19.39), kernel_min_us=4.52)              # a tick shorter than its fastest kernel + boundary
    assert not bad["consistent"] and bench.floors_inconsistent({"latency_floor": bad})
    plain = bench.latency_floor(6.7, 1.0, 10.0)                                                        # no same-pass figures: priced against the tick given
    assert plain["frac"] == [round(8.15 / 10.0, 3), round(8.6 / 10.0, 3)] and "frac_same_pass" not in plain
