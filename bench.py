#!/usr/bin/env python
"""bench.py -- rollback-resim entity-frames/sec on the stress_test workload (BASELINE.json).

One "step" = one SyncTest tick at depth D = 8 over N_ENT = 1M particles x 3 registered
components (Transform 40 B + Velocity 12 B + Ttl 8 B = 60 B/entity):
    [LoadGameState(F-8), Advance, (SaveGameState, Advance) x 7, SaveGameState(F), Advance]
  = 1 LoadWorld + 8 SaveWorld(+checksum) + 9 AdvanceWorld          
handed to libggrs_hip.so as ONE request list per tick (the C ABI's handle_requests boundary).  The
world is resident in HBM before the timed region; per tick the host sends ~1 KB of kernel arguments
and receives the 8 x 16 B Checksum(u128)s that the reference's `cell.save(frame, None, Some(checksum))`
needs.  Default host API: ggrs_hip_enqueue_requests / ggrs_hip_collect_checksums with one tick in
flight (a shim collects right before the next advance_frame()); `--sync` blocks on every tick.

  value     = entities x 9 advances x steps / seconds      (whole job, all ranks)
  roofline  = dominant kernel ggrs_jit_tick (the fused request group: read one snapshot, write D snapshots,
              write live once): the bytes the library counts per launch / its mean duration
              from HIP events riding on the dispatch inside libggrs_hip.so; `traffic` = HBM
              bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (profiles/roofline_traffic.json).
  extra_configs = (default N = 1 headline run only) BASELINE configs 2 / 4 / 5 (+ its spawning variant) and the all-columns-hot
              world, each measured in this process AFTER the headline's clock has stopped, each with its own in-run oracle
              parity; configs 2 / 4 also through a C loop (benches/tick_loop.c) next to this file's Python loop.
              SURVEY 8d's one-kernel-per-request figure (1656 B/entity-tick) is reported as
              per_request_equiv_*; `--no-groups` measures that path itself.
  cpu_baseline = the oracle's REFERENCE-SHAPED variant (kind "port"), 1 thread, bounded sample.

N > 1 (one process per GPU, launched by torch.distributed.run): speculative fan-out -- rank 0's
confirmed snapshot is broadcast ONCE over RCCL/xGMI; per step every rank runs ONE request list of the
same shape as the N = 1 tick (1 load + D saves + D+1 advances: the confirmed input for frame C, then its
own predicted-input branch for the following frames), two steps in flight, and ONE ncclAllGather of the checksums per
desync-detection interval (10 steps, the reference stress_test's default) on a side stream -- broadcast and all-gather are issued inside libggrs_hip.so (ggrs_hip_fanout_*); torch.distributed only
carries the ncclUniqueId and the timing barrier.  Weak scaling (per-GPU work fixed).
`--fanout` forces this code path at world size 1 (validation on a 1-GPU box).
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ADV_BYTES = 64                 # AdvanceWorld as its own kernel: r/w translation 12 + velocity 12 + ttl 8


def dbg_hooks(w):
    """A/B switches of the library's test hooks (the experiment behind profiles/r05h; none is set in a bench line that counts)."""
    if os.environ.get("BENCH_NO_LAZY_LIVE") == "1": w._lib.ggrs_dbg_set_lazy_live(w._p, 0)
    if os.environ.get("BENCH_VALUE_TAGS") in ("0", "1"): w._lib.ggrs_dbg_set_value_tags(w._p, int(os.environ["BENCH_VALUE_TAGS"]))    # A/B of the value-tag policy (default: by size)


def build_world(bg, cm, n, depth, stream=0, flags=0, checksum=True, schema="headline"):
    w = bg.World(n, max_depth=depth + 1, stream=stream, flags=flags)
    dbg_hooks(w)
    ids = cm.build_particles(w, checksum=checksum, schema=schema)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(depth + 1)
    w.set_synctest_check_distance(depth)
    return w, ids


def alu_view(diffuses_per_launch, launch_s, depth=8):
    """The dominant kernel priced against the chip's OTHER ceiling: SeaHash `diffuse` (two u64 multiplies) per second against what
    scripts/ubench_alu.hip measured on every CU (profiles/alu_ceiling.json).  Informational next to `roofline` (the kernel is bound by
    HBM): it says how far the hashing is from becoming the bound.  Never raises: a missing ceiling file gives frac None."""
    try:
        ceil = json.load(open(os.path.join(ROOT, "profiles", "alu_ceiling.json")))
        peak = float(ceil["diffuse_G_per_s"])
    except Exception:
        ceil, peak = {}, None
    ach = diffuses_per_launch / launch_s / 1e9 if launch_s > 0 else 0.0
    out = {"bound": "valu-int (u64 multiply)", "achieved": ach, "unit": "G diffuse/s", "peak": peak, "frac": (ach / peak) if peak else None,
           "peak_source": ceil.get("source"),
           "note": "algorithmic count (SURVEY 8d): 6 diffuse per entity per checksummed component per SaveWorld; the kernel executes fewer (hoisted order hash, memoised tails)"}
    out.update(alu_executed(out["frac"], depth))
    return out


def alu_executed(frac_algorithmic, depth, components=2):
    """What the generated kernel EXECUTES of the algorithmic 6 diffuses per entity, checksummed component and SaveWorld (stress_test: Transform.translation and Velocity,
    12 bytes each): 4 -- the full word, the finish of the inner hash, write_u64(inner) and the finish of the pair; the tail word's diffuse is memoised while the value stays
    (z never changes in this workload) and the order hash diffuse(K0 ^ order) is computed once per entity and launch.  `frac` counts the algorithm (and can exceed 1: the
    ceiling is diffuses per second); `frac_executed` is the share of the multiplier the launch really keeps busy with hashing."""
    if frac_algorithmic is None or not depth: return {}
    ratio = (4.0 * components * depth + 1.0) / (6.0 * components * depth)
    return {"executed_over_algorithmic": round(ratio, 4), "frac_executed": frac_algorithmic * ratio}


def rss_mb():
    try:
        with open("/proc/self/statm") as f: return round(int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20, 1)
    except Exception: return None


def tick_requests(bg, w, depth):
    """Steady-state SyncTest tick as a reusable ctypes array; frames patched per tick."""
    reqs = [bg.LoadGameState(0), bg.AdvanceFrame((0,))]
    for _ in range(depth - 1):
        reqs += [bg.SaveGameState(0), bg.AdvanceFrame((0,))]
    reqs += [bg.SaveGameState(0), bg.AdvanceFrame((0,))]
    arr, keep, n_save = w.build_requests(reqs)
    out = (C.c_uint64 * (2 * n_save))()
    save_idx = [i for i, r in enumerate(reqs) if isinstance(r, bg.SaveGameState)]

    # the frames of the Load and of the Saves are the only fields that change from tick to tick: written through ONE strided numpy view of the
    # ctypes array (a per-field ctypes store costs ~0.3 us each; at 10 k entities the tick is bound by exactly this kind of host work)
    fv = np.ndarray((len(reqs),), dtype=np.int32, buffer=arr, offset=type(arr[0]).frame.offset, strides=(C.sizeof(arr[0]),))
    idx = np.array([0] + save_idx, dtype=np.intp)
    rel = np.array([-depth] + [-depth + 1 + k for k in range(len(save_idx))], dtype=np.int32)

    def patch(F):
        fv[idx] = rel + np.int32(F)

    def run(F):
        patch(F)
        w.handle_requests_raw(arr, len(reqs), out)
        return out

    def enqueue(F):
        """ggrs_hip_enqueue_requests: the request list is copied into kernel arguments at enqueue time,
        so the same ctypes array can be re-patched for the next tick while this one runs."""
        patch(F)
        w.enqueue_requests_raw(arr, len(reqs))

    def collect():
        w.collect_checksums_raw(out, n_save)
        return out
    run.enqueue, run.collect = enqueue, collect
    return run, keep


def warm_ring(bg, w, depth):
    """Frames 0..depth: plain Save+Advance ticks so that Load(F-depth) has a snapshot."""
    for _ in range(depth + 1):
        w.handle_requests([bg.SaveGameState(w.frame), bg.AdvanceFrame((0,))])


def cpu_baseline_and_parity(n, depth, budget_ticks, frames_before_timed, gpu_cs, parity_ticks, schema="headline", time_refshaped=True):
    """The CPU path timed beside the GPU line, and the parity check of the same run.

    * reference-shaped oracle (per-save hash-map rebuild, per-entity lookups: the reference's cost structure) on ONE
      thread -- SaveWorld / LoadWorld are sequential `for` loops in the reference and AdvanceWorld runs single-threaded
      (src/lib.rs:236-240) -- and on 3 threads, one per registered component (the reference's per-component save / load /
      checksum systems may run on separate Bevy workers);
    * the oracle's FLAT variant (SoA + memcpy ring, the best a CPU port could do) on the host's cores, REPLAYING the very
      frames the timed GPU ticks covered: its Checksum(u128) of every SaveWorld must equal what the GPU returned for the
      first `parity_ticks` timed ticks (gpu_cs[k] = the D checksums of timed tick k)."""
    from oracle.binding import FLAT, REFSHAPED, OracleWorld, lib
    import common as cm

    def world(mode):
        w = OracleWorld(n, depth + 1, mode)
        ids = cm.build_particles(w, schema=schema)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
        w.set_depth(depth + 1)
        return w
    cores = os.cpu_count() or 1
    ef = n * (depth + 1)                       # entity-frames per tick
    secs = secs3 = None
    if time_refshaped:                         # (extra_configs: parity only -- the CPU timing legs belong to the headline)
        w = world(REFSHAPED)
        lib.gor_set_num_threads(1)
        secs = w.bench_synctest(depth, depth + 1, budget_ticks)
        lib.gor_set_ref_component_threads(3)
        lib.gor_set_num_threads(3)
        secs3 = w.bench_synctest(depth, 0, budget_ticks)
        lib.gor_set_ref_component_threads(1)
        del w
    # ---- flat port == parity replay: ring warm-up (D + 1 plain ticks) + the bench's warm-up ticks + P timed ticks
    P = max(0, min(parity_ticks, len(gpu_cs)))
    flat_threads = max(1, min(64, cores))
    wf = world(FLAT)
    lib.gor_set_num_threads(flat_threads)
    # frames_before_timed = ring warm-up + warm-up ticks + pre-heat ticks: the checker fast-forwards over them (advance-only,
    # then d + 1 plain [Save, Advance] ticks for the ring; gor_replay_synctest_from) and runs the P timed ticks for real
    fsecs, ocs = wf.replay_synctest_from(depth, frames_before_timed - (depth + 1), max(P, 1))
    del wf
    parity = {"checked_ticks": P, "checked_saves": P * depth, "equal": None}
    if P:
        want = ocs[len(ocs) - P * depth:]
        got = [c for tick in gpu_cs[:P] for c in tick]
        bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        parity["equal"] = (len(got) == len(want) == P * depth) and not bad
        parity["oracle"] = "oracle/ggrs_oracle.cpp FLAT variant, same seeded inputs, same request sequence"
        if bad:
            parity["first_mismatch"] = {"timed_tick": bad[0] // depth, "save": bad[0] % depth, "gpu": hex(got[bad[0]]), "oracle": hex(want[bad[0]])}
    flat = {"value": ef * max(P, 1) / fsecs, "unit": "entity-frames/s", "cores": flat_threads,
            "sample": f"{max(P, 1)} steady-state ticks of the oracle's flat SoA + memcpy-ring variant (OpenMP; the parity replay itself), {fsecs:.1f} s"}
    # 64 threads, not all cores: a 256-thread OpenMP team over this memcpy-bound port measured 2.1 M entity-frames/s against
    # 48.7 M at 64 threads on the 256-core gpurun host (profiles/r02a/bench.json) -- more threads than memory channels hurt
    flat["threads_note"] = f"{flat_threads} of {cores} host cores: measured faster than a {cores}-thread team (profiles/r02a)"
    lib.gor_set_num_threads(1)
    if not time_refshaped: return None, parity
    base = {"value": ef * budget_ticks / secs, "unit": "entity-frames/s", "cores": 1,
            "kind": "port",
            "sample": f"{budget_ticks} steady-state SyncTest ticks (depth {depth}) of the same {n}-entity x 3-component "
                      f"world on the oracle's reference-shaped storage (per-save HashMap rebuild), {secs:.1f} s",
            "host_cores_available": cores,
            "ref_shaped_3_threads": {"value": ef * budget_ticks / secs3, "unit": "entity-frames/s", "cores": 3,
                                     "sample": f"the same {budget_ticks} ticks with one thread per registered component for the per-component "
                                               f"checksum / save / load systems (AdvanceWorld and the entity systems stay sequential, src/lib.rs:236-240), {secs3:.1f} s"},
            "flat_soa_port": flat}
    return base, parity


def fanout_parity(n, depth, c_timed, raw, size, bpr, branch_input, confirmed_input, threads, spawn_rate=0):
    """N > 1 / --fanout parity gate: the gathered Checksum(u128) table of the first timed steps against ONE process walking every
    branch of every rank on the CPU oracle (tests/test_fanout_gloo.py::_serial_reference is the same recipe).  `raw` = what
    SpeculativeFanout kept: [(C, (size, bpr * depth, 2) u64)] for consecutive confirmed frames C = c_timed, c_timed + 1, ...
    The checker reaches frame c_timed by plain AdvanceWorlds with the confirmed inputs (what the steps before the timed
    region left behind, by determinism), saves it, and then runs for every step and every branch b of every rank
        [Load(C), Advance(confirmed input), Save(C+1), (Advance(predicted input of b), Save) x (depth-1), Advance(predicted)]."""
    from oracle.binding import FLAT, OracleWorld, lib
    import bevy_ggrs_amd as bg
    import common as cm
    lib.gor_set_num_threads(threads)
    t0 = time.perf_counter()
    cap = n + (2 * spawn_rate * (depth + 2) if spawn_rate else 0)
    o = OracleWorld(cap, depth + 1, FLAT)
    ids = cm.build_particles(o, with_spawn=bool(spawn_rate))
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(o, ids, n, vel, ttl)
    o.set_depth(depth + 1)
    spawn_fn = cm.frame_spawn_fn(spawn_rate) if spawn_rate else None

    def adv(frame, inp):
        a = bg.AdvanceFrame((inp,))
        if spawn_fn is not None and (inp & cm.INPUT_SPAWN): a.spawn_vx, a.spawn_vy = spawn_fn(frame)
        return a
    for f in range(c_timed):
        assert not (spawn_rate and confirmed_input(f) & cm.INPUT_SPAWN), "the bench's confirmed inputs never spawn"
        o.advance((confirmed_input(f),))
    assert o.frame == c_timed, (o.frame, c_timed)
    o.set_confirmed(c_timed)
    o.handle_requests([bg.SaveGameState(c_timed)])
    out = {"checked_steps": len(raw), "checked_branches": size * bpr, "checked_saves": 0, "equal": None,
           "oracle": "oracle/ggrs_oracle.cpp FLAT variant: one process walks every branch of every rank (serial reference), same seeded inputs"}
    bad = None
    for k, (C, table) in enumerate(raw):
        if C != c_timed + k:
            bad = {"step": k, "why": f"gathered step carries confirmed frame {C}, expected {c_timed + k}"}; break
        o.set_confirmed(C)
        for r in range(size):
            for j in range(bpr):
                b = r * bpr + j
                reqs = [bg.LoadGameState(C), adv(C, confirmed_input(C)), bg.SaveGameState(C + 1)]
                for i in range(1, depth):
                    reqs += [adv(C + i, branch_input(b, C + i)), bg.SaveGameState(C + 1 + i)]
                reqs.append(adv(C + depth, branch_input(b, C + depth)))
                want = o.handle_requests(reqs)
                got = [int(table[r, j * depth + i, 0]) | (int(table[r, j * depth + i, 1]) << 64) for i in range(depth)]
                out["checked_saves"] += depth
                if got != want and bad is None:
                    i = next(i for i in range(depth) if got[i] != want[i])
                    bad = {"step": k, "rank": r, "branch": b, "save": i, "gpu": hex(got[i]), "oracle": hex(want[i])}
        if bad: break
    out["equal"] = bad is None and len(raw) > 0
    if bad: out["first_mismatch"] = bad
    out["oracle_seconds"] = round(time.perf_counter() - t0, 2)
    lib.gor_set_num_threads(1)
    return out


def measure_traffic_now(args, budget_s=120.0):
    """HBM bytes per `ggrs_jit_tick` launch, MEASURED in this run: bench.py starts `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two do not fit
    one pass on gfx950, MI355X_MICROARCH.md; counters only, no trace domains) around its own short form (76 ticks, no pre-heat, no CPU legs, no extras) and reads the
    counter CSVs: KiB per dispatch, second half of the dispatches (steady state).  The generated kernel reads 4 bytes per lane: FETCH_SIZE is taken as reported (the
    guide's x2 correction is calibrated for 16-byte-per-lane reads), WRITE_SIZE is calibrated against a known snapshot copy.  Any failure leaves the figure to the
    committed profile and says why."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe): return {"error": "rocprofv3 not found"}
    t0 = time.perf_counter()
    out = {"passes": {}, "command": "rocprofv3 --pmc <counter> -f csv -- python bench.py --steps 60 --warmup 16 --no-cpu-baseline --preheat-ms 0 --no-extra --no-traffic"}
    tmp = tempfile.mkdtemp(prefix="ggrs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"): env.pop(key, None)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            if time.perf_counter() - t0 > budget_s: out["passes"][counter] = {"skipped": "budget"}; continue
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-f", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.abspath(__file__), "--steps", "60", "--warmup", "16", "--no-cpu-baseline",
                   "--preheat-ms", "0", "--no-extra", "--no-traffic", "--entities", str(args.entities), "--depth", str(args.depth), "--schema", args.schema]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=max(30.0, budget_s - (time.perf_counter() - t0)))
            except subprocess.TimeoutExpired:
                out["passes"][counter] = {"error": "timeout"}; continue
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and "ggrs_jit_tick" in row.get("Kernel_Name", ""): vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 20:
                out["passes"][counter] = {"error": f"rc {r.returncode}, {len(vals)} dispatches", "stderr_tail": r.stderr[-300:]}; continue
            half = vals[len(vals) // 2:]
            out["passes"][counter] = {"KiB_per_dispatch_mean": sum(half) / len(half), "dispatches": len(vals)}
        f_, w_ = out["passes"].get("FETCH_SIZE", {}), out["passes"].get("WRITE_SIZE", {})
        if "KiB_per_dispatch_mean" in f_ and "KiB_per_dispatch_mean" in w_:
            out["hbm_bytes_per_launch"] = (f_["KiB_per_dispatch_mean"] + w_["KiB_per_dispatch_mean"]) * 1024
            out["fetch_bytes"], out["write_bytes"] = f_["KiB_per_dispatch_mean"] * 1024, w_["KiB_per_dispatch_mean"] * 1024
            out["source"] = ("MEASURED IN THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate counter-only passes of this command's short form, started by bench.py "
                             "after its clock stopped); FETCH as reported (4-byte-per-lane reads), WRITE calibrated (MI355X_MICROARCH.md)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def oracle_checksum_at(n, depth, frame, confirmed_input, spawn_rate=0):
    """Checksum(u128) of SaveWorld at `frame` of ONE oracle world that simulated the confirmed inputs frame by frame from the synthetic start (what an adopted
    branch state must equal)."""
    from oracle.binding import FLAT, OracleWorld, lib
    import common as cm
    lib.gor_set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    o = OracleWorld(n + (2 * spawn_rate * (depth + 2) if spawn_rate else 0), depth + 1, FLAT)
    ids = cm.build_particles(o, with_spawn=bool(spawn_rate))
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(o, ids, n, vel, ttl)
    for f in range(frame):
        o.advance((confirmed_input(f),))
    o.set_depth(2)
    cs = o.save()
    lib.gor_set_num_threads(1)
    return cs


def latency_floor(kernel_us, launches_per_tick, tick_us, same_pass=None, kernel_min_us=None):
    """Small worlds (the whole ring lives in L2 / the Infinity Cache) are bound by launch latency, not by HBM: a tick cannot be shorter than
    its kernels plus one dependent same-stream boundary per launch -- 1.45 us between trivial kernels, 1.7-1.9 us between real streaming
    ones (MI355X_MICROARCH.md, price list row `boundary`).  Reported next to `roofline` for BASELINE configs 2 and 4.
    same_pass = (kernel us per tick, tick us) measured in ONE pass -- HIP events riding on the dispatches of the very ticks whose wall time is taken: the kernels
    are inside those ticks, so the floor fraction cannot exceed 1; `frac` is priced there (a fraction above 1 means the model's inputs are inconsistent and FAILS
    the bench line: `consistent`).  The un-instrumented timed region's tick is reported next to it (`achieved_us_per_tick`): it runs without the timing events."""
    k_us, t_us = same_pass if same_pass else (kernel_us * launches_per_tick, tick_us)
    lo, hi = k_us + 1.45 * launches_per_tick, k_us + 1.9 * launches_per_tick
    out = {"bound": "launch latency (dependent same-stream kernel boundary)", "boundary_us": [1.45, 1.9], "source": "MI355X_MICROARCH.md, 'Persistent kernels: synchronisation and hand-off price list', row boundary",
           "kernel_us_per_tick": round(k_us, 3), "launches_per_tick": launches_per_tick, "floor_us_per_tick": [round(lo, 2), round(hi, 2)],
           "tick_us_same_pass": round(t_us, 3) if same_pass else None, "achieved_us_per_tick": tick_us,
           "priced": "kernel time and tick time from the SAME instrumented pass" if same_pass else "kernel time from an instrumented pass, tick from the timed region",
           "frac": [round(lo / t_us, 3), round(hi / t_us, 3)] if t_us else None}
    # the lower boundary figure is the floor: kernels + the shortest dependent hand-off the platform does; the upper one is what streaming kernels usually pay
    out["consistent"] = bool(t_us and lo <= t_us * 1.0005)
    if same_pass and tick_us:
        # the timing events themselves slow a small world's tick (the runtime timestamps every dispatch: +7..12 us per tick at 10 k..100 k), so the same-pass fraction
        # prices the INSTRUMENTED loop: it is kept as `frac_same_pass` (the consistency check), and `frac` is what can be said about the un-instrumented timed tick without
        # mixing passes -- a LOWER BOUND: the fastest kernel any pass has seen + one boundary cannot be longer than a tick that contains that kernel.  (The mean kernel of
        # the instrumented pass + a boundary is `frac_mixed_passes`: it can exceed 1 -- kernels run slower between timing events -- and is informational.)
        out["instrumentation_us_per_tick"] = round(t_us - tick_us, 2)
        out["frac_same_pass"] = out["frac"]
        out["frac_mixed_passes"] = [round(lo / tick_us, 3), round(hi / tick_us, 3)]
        if kernel_min_us:
            out["frac_timed_tick_lower_bound"] = round((kernel_min_us * launches_per_tick + 1.45 * launches_per_tick) / tick_us, 3)
            out["frac"] = out["frac_timed_tick_lower_bound"]
            out["frac_basis"] = "lower bound against the un-instrumented timed tick: (fastest kernel seen + 1.45 us boundary) x launches per tick / achieved_us_per_tick; frac_same_pass prices the instrumented loop"
            out["consistent"] = bool(out["consistent"] and out["frac"] <= 1.0005)
    return out


def platform_loop_floor(inflight, tick_us, kernel_us=None):
    """What a bare loop with one tick in flight costs on this platform for a kernel of this duration (scripts/ubench_launch: launch with a completion event, poll
    the event, launch the next -- no library, no request list, an empty kernel body that spins for the given time): 6.1 us per tick around a 2 us kernel, 8.1 around
    5.6 us, 9.0 around 8 us.  Interpolated at the instrumented kernel time.  Informational, from the committed measurement."""
    try:
        u = json.load(open(os.path.join(ROOT, "profiles", "r05j", "ubench_launch.json")))
        pts = [(2.0, u["pipelined_tick_kernel_2.0us_ext_event_query_spin"]), (5.6, u["pipelined_tick_kernel_5.6us_ext_event_query_spin"]), (8.0, u["pipelined_tick_kernel_8.0us_ext_event_query_spin"])]
    except Exception:
        return None
    if inflight != 1 or not kernel_us: return {"note": "measured for one tick in flight only", "source": "profiles/r05j/ubench_launch.json"}
    k = float(kernel_us)
    if k <= pts[0][0]: f = pts[0][1]
    elif k >= pts[-1][0]: f = pts[-1][1] + (k - pts[-1][0])
    else:
        (k0, f0), (k1, f1) = (pts[0], pts[1]) if k <= pts[1][0] else (pts[1], pts[2])
        f = f0 + (f1 - f0) * (k - k0) / (k1 - k0)
    return {"bare_loop_us_per_tick": round(f, 2), "at_kernel_us": round(k, 2), "achieved_us_per_tick": round(tick_us, 3), "frac": round(f / tick_us, 3) if tick_us else None,
            "source": "profiles/r05j/ubench_launch.json (scripts/ubench_launch, hipEventQuery spin: 2.0 / 5.6 / 8.0 us kernels -> 6.13 / 8.09 / 9.04 us per tick), interpolated"}


def read_clocks():
    """sclk / mclk / power right now: sysfs (pp_dpm_*: the starred level; hwmon power) when the container exposes it, else
    `rocm-smi`.  Telemetry for the JSON line only."""
    import glob
    out = {}
    try:
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            def star(name):
                try:
                    for ln in open(os.path.join(card, name)).read().splitlines():
                        if ln.rstrip().endswith("*"): return ln.split(":", 1)[1].replace("*", "").strip()
                except OSError:
                    return None
            sclk, mclk = star("pp_dpm_sclk"), star("pp_dpm_mclk")
            if sclk or mclk:
                out = {"source": "sysfs " + card, "sclk": sclk, "mclk": mclk}
                for pw in glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")):
                    try: out["power_w"] = int(open(pw).read()) / 1e6; break
                    except (OSError, ValueError): pass
                return out
    except Exception:
        pass
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=15)
        j = json.loads(r.stdout[r.stdout.index("{"):])
        c = j.get("card0", next(iter(j.values())))
        out = {"source": "rocm-smi", "sclk": c.get("sclk clock speed:"), "mclk": c.get("mclk clock speed:")}
        for k, v in c.items():
            if "ower" in k and "(W)" in k: out["power_w"] = v; break
    except Exception as e:       # noqa: BLE001 -- telemetry must never fail the bench
        out = {"source": "unavailable", "why": f"{type(e).__name__}: {e}"[:120]}
    return out


def spawn_ranks(args, argv):
    """`bench.py --gpus N` without a launcher around it: start the N ranks here, one process per GPU (rank r on HIP device r),
    over 127.0.0.1; rank 0's JSON line is this process's output.  Returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    if args.dry_run:
        ndev = None
    else:
        import torch
        ndev = torch.cuda.device_count()
        if ndev == 0:
            print("bench.py: no GPU visible", file=sys.stderr); return 2
        if n > ndev and not args.oversubscribe:
            print(f"bench.py: --gpus {n} but only {ndev} device(s) visible (pass --oversubscribe to share devices: correctness "
                  f"only, and RCCL builds that refuse two ranks per device will fail in ncclCommInitRank)", file=sys.stderr)
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmds = []
    for r in range(n):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        cmds.append((env, [sys.executable, os.path.abspath(__file__)] + argv))
    if args.dry_run:
        for env, cmd in cmds:
            print(json.dumps({"env": env, "cmd": cmd}))
        return 0
    procs = []
    for r, (env, cmd) in enumerate(cmds):
        procs.append(subprocess.Popen(cmd, env={**os.environ, **env}, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is None: continue
                pending.remove(p)
                if c != 0 and rc == 0:
                    rc = c
                    for q in pending: q.terminate()        # one rank failed: the others would wait in a collective forever
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None: p.kill()
    return rc


def measure_single(bg, cm, torch, args, contig=False, light=False):
    """One single-GPU measurement in this process: warm-up, pre-heat, K timed ticks, then the host-timeline pass and the HIP-event pass for the
    per-kernel roofline.  light: no telemetry, no checksum capture."""
    n, D, K, W = args.entities, args.depth, args.steps, args.warmup
    stream = torch.cuda.current_stream().cuda_stream
    flags = (bg.GGRS_WORLD_UNFUSED if args.unfused else 0) | (bg.GGRS_WORLD_NT_COPY if args.nt else 0) | (bg.GGRS_WORLD_NO_GROUPS if args.no_groups else 0)
    w, ids = build_world(bg, cm, n, D, stream=stream, flags=flags, checksum=not args.no_checksum, schema=args.schema)
    m = {}
    warm_ring(bg, w, D)
    run, _keep = tick_requests(bg, w, D)
    m["clocks_start"] = None if light else read_clocks()
    # the interpreter's cycle collector stays out of the timed region (profiles/r03zi: one generation-2 pass inside it reads as a 35 ms
    # tick); collected HERE, before warm-up and pre-heat, so that the device is not left idle for that long right before the timed ticks
    gc.collect(); gc.disable()
    for _ in range(W):
        run(w.frame)
    torch.cuda.synchronize()
    # ---- pre-heat: the SAME tick on the SAME world for a fixed wall time, reported on its own (not part of `warmup`): the
    # chip clocks up over the first tens of milliseconds of load, and a 20-step timed region is 2.4 ms long
    pre_t0 = time.perf_counter(); pre_n = 0
    if args.preheat_ms > 0:
        run.enqueue(w.frame); pre_n = 1
        while (time.perf_counter() - pre_t0) * 1e3 < args.preheat_ms:
            run.enqueue(w.frame); run.collect(); pre_n += 1
        run.collect()
        w.synchronize()
    # the library builds a kernel specialised for the tick's group shape on a worker thread once it has seen the shape 16 times
    # (include/ggrs_hip.h ggrs_hip_specialise_wait): like the generated kernel's own compile at seal this is set-up, not the timed
    # region -- wait for it, then let the new kernel run for a moment
    spec_ready = bool(w.specialise_wait()) if not args.no_specialise_wait else False
    if spec_ready:
        run.enqueue(w.frame); pre_n += 1
        for _ in range(199):
            run.enqueue(w.frame); run.collect(); pre_n += 1
        run.collect()
        w.synchronize()
    m["preheat"] = {"ms": (time.perf_counter() - pre_t0) * 1e3, "ticks": pre_n, "requested_ms": args.preheat_ms, "specialised_kernel_ready": spec_ready}
    m["frames_before_timed"] = w.frame
    rss0 = rss_mb()
    gpu_cs = []                                  # per timed tick: its D Checksum(u128)s, what cell.save() receives
    stamps = []
    take = (lambda out: None) if light else (lambda out: (gpu_cs.append(bytes(out)), stamps.append(time.perf_counter())))
    # the synchronize that opens the timed region is the LAST thing before the clock starts (the bookkeeping above used to sit between the two: a few
    # hundred microseconds of idle device, after which the first launch took 90-160 us instead of the kernel's 45: profiles/r05o)
    torch.cuda.synchronize()
    if args.sync:
        t0 = time.perf_counter()
        for _ in range(K):
            take(run(w.frame))
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
    else:
        # one tick in flight ahead of the host: enqueue tick k+1, then collect tick k's checksums
        # (what a shim does: cell.save() right before the next advance_frame()).  Every one of the K
        # ticks is enqueued AND collected inside the timed region.
        w.synchronize()
        t0 = time.perf_counter()
        run.enqueue(w.frame)
        for _ in range(K - 1):
            run.enqueue(w.frame)
            take(run.collect())
        take(run.collect())
        torch.cuda.synchronize()                 # device-wide: covers the world's stream (it IS torch's current stream); a second, per-stream wait here only cost a marker packet
        secs = time.perf_counter() - t0
    gc.enable()
    m["secs"] = secs
    m["clocks_end"] = None if light else read_clocks()
    m["rss_mb"] = [rss0, rss_mb()]               # resident set of this process before / after the timed region (soaks: the runtime must not grow with the tick count)
    m["live"] = w.active_count()
    m["gpu_cs"] = [[int.from_bytes(b[16 * k:16 * k + 16], "little") for k in range(D)] for b in gpu_cs]
    if stamps:
        d = [(b - a) * 1e6 for a, b in zip([t0] + stamps[:-1], stamps)]
        m["tick_wall_us"] = {"first5": [round(x, 1) for x in d[:5]], "last5": [round(x, 1) for x in d[-5:]], "median": round(sorted(d)[len(d) // 2], 1),
                             "worst3": [(i, round(d[i], 1)) for i in sorted(range(len(d)), key=lambda i: -d[i])[:3]],
                             "after_last_collect": round((t0 + secs - stamps[-1]) * 1e6, 1),      # the two synchronize calls that close the region
                             "note": "host interval between consecutive collects in the timed region (tick k+1 is already enqueued when tick k is collected)"}
    m["f_end"] = w.frame
    if not args.sync and not light:
        m["host_timeline"] = timeline_pass(w, run, min(K, 50))
    # ---- instrumented pass for the per-kernel roofline: HIP events riding on every dispatch (kernel begin / end, what rocprofv3's kernel
    # trace reports), through the SAME host API as the timed region -- pipelined ticks run back to back on a busy device with the previous
    # tick's first Save still in the caches, exactly the conditions of the timed ticks (the pass used to go through the
    # blocking API, where the device idles between ticks)
    w.profile_enable(True)
    n_prof = min(K, 50)
    w.synchronize()
    tp0 = time.perf_counter()
    if args.sync:
        for _ in range(n_prof):
            run(w.frame)
    else:
        run.enqueue(w.frame)
        for _ in range(n_prof - 1):
            run.enqueue(w.frame); run.collect()
        run.collect()
        w.synchronize()
    m["prof_pass_us"] = (time.perf_counter() - tp0) * 1e6 / max(n_prof, 1)        # tick time of the pass the kernel times come from
    m["prof"] = w.profile_read()
    m["prof_bytes"] = w.profile_bytes()
    tick_us = w.profile_launches("tick")
    w.profile_enable(False)
    if tick_us:
        srt = sorted(tick_us)
        m["launch_us"] = {"first5": [round(x, 2) for x in tick_us[:5]], "last5": [round(x, 2) for x in tick_us[-5:]],
                          "min": round(srt[0], 2), "median": round(srt[len(srt) // 2], 2), "max": round(srt[-1], 2), "n": len(tick_us)}
    m["info"] = w.kernel_info()
    w.close()
    return m


def measure_p2p(bg, cm, torch, args):
    """BASELINE config 4: a 2-player P2P session's rollbacks at 120 ms RTT, 100 k entities -- no sockets: every tick a late remote
    input invalidates the last r predicted frames (r drawn per tick from 0..depth, as tests/common.py::P2PShapeDriver draws it), so
    the request list is [Load(F-r), Adv, (Save, Adv) x (r-1)] + [Save(F), Adv]; ConfirmedFrameCount trails by `depth` frames.
    One list per tick through enqueue / collect with one tick in flight; every Checksum(u128) is compared with the CPU oracle
    driven by the same script."""
    n, R, K, W = args.entities, args.depth, args.steps, args.warmup
    stream = torch.cuda.current_stream().cuda_stream
    w = bg.World(n, max_depth=R + 1, stream=stream)
    dbg_hooks(w)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(R); w.set_synctest_check_distance(-1)
    lists = {}
    for r in range(R):                                        # r frames rolled back: one reusable ctypes list per depth
        reqs = ([bg.LoadGameState(0)] + [x for i in range(r) for x in (([bg.SaveGameState(0)] if i else []) + [bg.AdvanceFrame((0,))])]) if r else []
        reqs += [bg.SaveGameState(0), bg.AdvanceFrame((0,))]
        arr, keep, n_save = w.build_requests(reqs)
        lists[r] = (arr, keep, n_save, len(reqs), [i for i, q in enumerate(reqs) if isinstance(q, bg.SaveGameState)])
    rng = np.random.default_rng(4)
    out = (C.c_uint64 * (2 * R))()
    F, script, got = 0, [], []

    def enqueue():
        nonlocal F
        r = int(rng.integers(0, R + 1)); r = min(r, F, R - 1)
        arr, _k, n_save, n_req, save_idx = lists[r]
        if r: arr[0].frame = F - r
        for k, i in enumerate(save_idx): arr[i].frame = F - r + (k + 1 if r else 0) if r else F
        if r: arr[save_idx[-1]].frame = F
        if F - R >= 0: w.set_confirmed(F - R)
        w.enqueue_requests_raw(arr, n_req)
        script.append((F, r, n_save)); F += 1

    def collect(k):
        n_save = script[k][2]
        w.collect_checksums_raw(out, n_save)
        got.append(bytes(out)[:16 * n_save])                  # decoded after the run: no Python objects per Save inside the timed region
    gc.collect(); gc.disable()               # a generation-2 pass of the interpreter's collector inside a 2.5 ms region reads as a 35 ms tick (profiles/r03zi)
    for _ in range(W):
        enqueue(); collect(len(got))
    # the library gives every group shape it has seen 16 times a kernel of its own, built one at a time on a worker thread
    # (include/ggrs_hip.h ggrs_hip_specialise_wait); a P2P session has one shape per rollback length.  Set-up like the compile at
    # seal, not the timed region: keep the session running until no new kernel appears (every tick goes to the oracle check too)
    settle = {"ticks": 0, "rounds": 0}
    if not args.no_specialise_wait:
        last = None
        for _ in range(24):
            for _ in range(60):
                enqueue(); collect(len(got))
            w.specialise_wait()
            settle["ticks"] += 60; settle["rounds"] += 1
            now = w.kernel_info().get("specialised_kernel")
            if settle["ticks"] >= 360 and now == last and "building" not in str(now): break      # 360 ticks: every length has come up 16 times
            last = now
    w.synchronize(); torch.cuda.synchronize()
    first = len(got)
    stamps = []
    t0 = time.perf_counter()
    enqueue()
    for _ in range(K - 1):
        enqueue(); collect(len(got)); stamps.append(time.perf_counter())
    collect(len(got)); stamps.append(time.perf_counter())
    w.synchronize(); torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    gc.enable()
    gaps = [(b - a) * 1e6 for a, b in zip([t0] + stamps[:-1], stamps)]
    tick_wall = {"median": round(sorted(gaps)[len(gaps) // 2], 1), "first5": [round(x, 1) for x in gaps[:5]],
                 "worst5": [(i, script[first + i][1], round(gaps[i], 1)) for i in sorted(range(len(gaps)), key=lambda i: -gaps[i])[:5]],
                 "note": "host interval between consecutive collects in the timed region; worst5 = (tick, rollback length, us)"}
    advances = sum(r + 1 for _f, r, _s in script[first:])
    w.profile_enable(True)
    w.synchronize()
    tp0 = time.perf_counter()
    for _ in range(min(K, 50)):
        enqueue(); collect(len(got))
    w.synchronize()
    prof_pass_us = (time.perf_counter() - tp0) * 1e6 / max(min(K, 50), 1)      # tick time of the pass the kernel times come from
    prof, pbytes, info = w.profile_read(), w.profile_bytes(), w.kernel_info()
    w.profile_enable(False)
    live = w.active_count(); w.close()
    got = [[int.from_bytes(b[16 * i:16 * i + 16], "little") for i in range(len(b) // 16)] for b in got]
    # ---- the CPU oracle under the same script
    from oracle.binding import FLAT, OracleWorld, lib as olib0
    olib0.gor_set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    o = OracleWorld(n, R + 1, FLAT)
    oids = cm.build_particles(o); cm.spawn_particles(o, oids, n, vel, ttl); o.set_depth(R)
    want = []
    for (f, r, _s) in script:
        reqs = ([bg.LoadGameState(f - r)] + [x for i in range(r) for x in (([bg.SaveGameState(f - r + i)] if i else []) + [bg.AdvanceFrame((0,))])]) if r else []
        reqs += [bg.SaveGameState(f), bg.AdvanceFrame((0,))]
        if f - R >= 0: o.set_confirmed(f - R)
        want.append(o.handle_requests(reqs))
    olib0.gor_set_num_threads(1)
    # ---- the CPU path beside it: the oracle's reference-shaped storage, one thread, the first ticks of the same script
    cpu = None
    if not args.no_cpu_baseline:
        from oracle.binding import REFSHAPED, lib as olib
        olib.gor_set_num_threads(1)
        r_ = OracleWorld(n, R + 1, REFSHAPED)
        rids = cm.build_particles(r_); cm.spawn_particles(r_, rids, n, vel, ttl); r_.set_depth(R)
        sample, adv_n = script[:max(8, 8 * args.cpu_ticks)], 0
        tc = time.perf_counter()
        for (f, r, _s) in sample:
            reqs = ([bg.LoadGameState(f - r)] + [x for i in range(r) for x in (([bg.SaveGameState(f - r + i)] if i else []) + [bg.AdvanceFrame((0,))])]) if r else []
            reqs += [bg.SaveGameState(f), bg.AdvanceFrame((0,))]
            if f - R >= 0: r_.set_confirmed(f - R)
            r_.handle_requests(reqs); adv_n += r + 1
        tc = time.perf_counter() - tc
        cpu = {"value": n * adv_n / tc, "unit": "entity-frames/s", "cores": 1, "kind": "port", "host_cores_available": os.cpu_count(),
               "sample": f"the first {len(sample)} ticks of the same rollback script ({adv_n} AdvanceWorlds) on the oracle's reference-shaped storage (per-save HashMap rebuild), {tc:.1f} s"}
    return {"secs": secs, "advances": advances, "live": live, "prof": prof, "prof_bytes": pbytes, "info": info, "cpu_baseline": cpu, "prof_pass_us": prof_pass_us,
            "parity": {"checked_ticks": len(script), "checked_saves": sum(len(x) for x in want), "equal": got == want,
                       "oracle": "oracle/ggrs_oracle.cpp FLAT variant driven by the same rollback script"},
            "mean_rollback": sum(r for _f, r, _s in script[first:first + K]) / K, "settle": settle, "tick_wall_us": tick_wall}


def timeline_pass(w, run, ticks):
    """Where the HOST spends a tick (ggrs_hip_host_timeline), measured over `ticks` pipelined ticks after the timed region -- per tick, microseconds."""
    w.synchronize()
    w.host_timeline(1)
    run.enqueue(w.frame)
    for _ in range(ticks - 1):
        run.enqueue(w.frame); run.collect()
    run.collect()
    t = w.host_timeline(0)
    n = max(1, t["collect_calls"])
    per = {k: round(v / n, 3) for k, v in t.items() if k.endswith("_us")}
    per["enqueue_bookkeeping_us"] = round(per["enqueue_us"] - per["validate_us"] - per["launch_call_us"], 3)
    per["ticks"] = n; per["launches_per_tick"] = round(t["launches"] / n, 3)
    per["note"] = ("library time inside ggrs_hip_enqueue_requests (validation + group bookkeeping + the HIP launch call) and ggrs_hip_collect_checksums (batch event + fold-forward tags + "
                   "hashing) per tick; what the bench's own loop adds on top is the difference to tick_wall_us")
    return per


_TICK_LOOP = None


def tick_loop_lib():
    """benches/libtick_loop.so: the bench's ticks driven from C through the public C ABI (benches/tick_loop.c)."""
    global _TICK_LOOP
    if _TICK_LOOP is None:
        lib = C.CDLL(os.path.join(ROOT, "benches", "libtick_loop.so"))
        P = C.c_void_p
        lib.ggrs_bench_synctest_loop.restype = C.c_int
        lib.ggrs_bench_synctest_loop.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.ggrs_bench_p2p_loop.restype = C.c_int
        lib.ggrs_bench_p2p_loop.argtypes = [P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _TICK_LOOP = lib
    return _TICK_LOOP


def c_loop_synctest(bg, cm, torch, n, D, K, schema="headline", inflight=1, kernel_us=None, parity_ticks=8):
    """The SyncTest tick of `measure_single` with NO Python between enqueue and collect: benches/tick_loop.c drives the C ABI (what a Rust / C++ shim costs).
    Own world, ring warm-up, 400 untimed ticks (the specialised kernel comes in), K timed ticks whose checksums are all kept; the first `parity_ticks` timed
    ticks are replayed on the CPU oracle."""
    lib = tick_loop_lib()
    stream = torch.cuda.current_stream().cuda_stream
    w, _ids = build_world(bg, cm, n, D, stream=stream, schema=schema)
    warm_ring(bg, w, D)
    secs = C.c_double(0)
    warm = 400
    rc = lib.ggrs_bench_synctest_loop(w._p, D, warm // 2, inflight, None, C.byref(secs), None); assert rc == 0, rc
    w.specialise_wait()
    rc = lib.ggrs_bench_synctest_loop(w._p, D, warm // 2, inflight, None, C.byref(secs), None); assert rc == 0, rc
    frames_before = w.frame
    cs = (C.c_uint64 * (2 * D * K))(); tick_us = (C.c_double * K)()
    gc.collect(); gc.disable()
    rc = lib.ggrs_bench_synctest_loop(w._p, D, K, inflight, cs, C.byref(secs), tick_us)
    gc.enable()
    assert rc == 0, rc
    # the kernel's own duration UNDER THIS LOOP (HIP events riding on the dispatches of 200 more ticks): a small world's kernel is only as fast as the rocprofv3
    # trace says (5.6 us at 10 k) while the GPU is kept busy -- behind a host-bound loop it starts from an idle, clock-gated chip and reads 8 us
    w.profile_enable(True)
    s2 = C.c_double(0); n_prof = 200                       # (the library's pool holds timing events for 256 launches: beyond it every launch creates two)
    rc = lib.ggrs_bench_synctest_loop(w._p, D, n_prof, inflight, None, C.byref(s2), None); assert rc == 0, rc
    lus_raw = w.profile_launches("tick"); lus = sorted(lus_raw); w.profile_enable(False)
    k_us = lus[len(lus) // 2] if lus else (kernel_us or 0.0)
    same = (sum(lus_raw) / n_prof, s2.value / n_prof * 1e6) if lus_raw else None      # kernel us per tick and tick us of the SAME pass
    live = w.active_count(); w.close()
    t = sorted(tick_us)
    out = {"host_loop": "C (benches/tick_loop.c through the C ABI)", "ticks_in_flight": inflight, "steps": K, "ms_per_step": secs.value / K * 1e3, "value": live * (D + 1) * K / secs.value, "unit": "entity-frames/s",
           "tick_wall_us": {"median": round(t[K // 2], 2), "p10": round(t[K // 10], 2), "p90": round(t[(9 * K) // 10], 2), "first": round(tick_us[0], 2)},
           "kernel_us": {"median_under_this_loop": round(k_us, 2), "min": round(lus[0], 2) if lus else None, "launches": len(lus), "under_the_python_loop": kernel_us}}
    # kernel time: the median of the instrumented pass (timing events ride on the dispatches and come from a pool since profiles/r05j -- creating them per launch
    # kept the host behind the device and the kernels started from an idle chip: 8.5 instead of 5.3 us); tick: the un-instrumented timed region
    if k_us: out["latency_floor"] = latency_floor(k_us, 1.0, secs.value / K * 1e6, same_pass=same if inflight == 1 else None, kernel_min_us=lus[0] if lus else None)
    out["platform_floor"] = platform_loop_floor(inflight, secs.value / K * 1e6, k_us)
    P = min(parity_ticks, K)
    if P:
        from oracle.binding import FLAT, OracleWorld, lib as olib
        o = OracleWorld(n, D + 1, FLAT)
        ids = cm.build_particles(o, schema=schema)
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(o, ids, n, vel, ttl); o.set_depth(D + 1)
        olib.gor_set_num_threads(max(1, min(64, os.cpu_count() or 1)))
        _s, ocs = o.replay_synctest_from(D, frames_before - (D + 1), P)
        olib.gor_set_num_threads(1)
        got = [int(cs[2 * i]) | (int(cs[2 * i + 1]) << 64) for i in range(P * D)]
        out["parity"] = {"checked_ticks": P, "checked_saves": P * D, "equal": got == ocs[len(ocs) - P * D:]}
    return out


def c_loop_p2p(bg, cm, torch, n, R, K, kernel_us=None, launches_per_tick=1.0):
    """BASELINE config 4 through the C loop: the same rollback script generator as measure_p2p, one list per tick, one tick in flight; every Save checked against the oracle."""
    lib = tick_loop_lib()
    stream = torch.cuda.current_stream().cuda_stream
    w = bg.World(n, max_depth=R + 1, stream=stream)
    dbg_hooks(w)
    ids = cm.build_particles(w)
    vel, ttl = cm.synthetic_particles(n, ttl="throughput")
    cm.spawn_particles(w, ids, n, vel, ttl)
    w.set_depth(R); w.set_synctest_check_distance(-1)
    rng = np.random.default_rng(4)
    warm, n_prof = 480, 200
    total = warm + K + n_prof
    rl = np.zeros(total, dtype=np.uint8)
    for F in range(total): rl[F] = min(int(rng.integers(0, R + 1)), F, R - 1)
    cs = (C.c_uint64 * (2 * R * total))(); ncs = (C.c_uint32 * total)(); secs = C.c_double(0); tick_us = (C.c_double * K)()
    rlp = rl.ctypes.data_as(C.POINTER(C.c_uint8))
    off = lambda k: (C.cast(C.byref(cs, 16 * R * k), C.POINTER(C.c_uint64)), C.cast(C.byref(ncs, 4 * k), C.POINTER(C.c_uint32)), C.cast(C.byref(rl.ctypes.data_as(C.POINTER(C.c_uint8)).contents, k), C.POINTER(C.c_uint8)))
    # warm-up in rounds of 40 ticks: every rollback length gets its own specialised kernel after 16 sightings, built one at a time on a worker thread
    # (measure_p2p's settle loop; the code objects are on disk by now, this world only has to see the shapes and load them)
    done = 0
    while done < warm:
        cc, nn, rr = off(done)
        rc = lib.ggrs_bench_p2p_loop(w._p, R, 40, rr, cc, nn, C.byref(secs), None); assert rc == 0, rc
        w.specialise_wait()
        done += 40
    c3, n3, r3 = off(warm)
    gc.collect(); gc.disable()
    rc = lib.ggrs_bench_p2p_loop(w._p, R, K, r3, c3, n3, C.byref(secs), tick_us)
    gc.enable()
    assert rc == 0, rc
    # the kernels' own durations under this loop (see c_loop_synctest): the script's next 200 ticks with HIP events riding on the dispatches
    w.profile_enable(True)
    c4, n4, r4 = off(warm + K); s2 = C.c_double(0)
    rc = lib.ggrs_bench_p2p_loop(w._p, R, n_prof, r4, c4, n4, C.byref(s2), None); assert rc == 0, rc
    lus = w.profile_launches("tick"); w.profile_enable(False)
    k_us = (sum(lus) / len(lus)) if lus else (kernel_us or 0.0)
    lpt = (len(lus) / n_prof) if lus else launches_per_tick
    same = (sum(lus) / n_prof, s2.value / n_prof * 1e6) if lus else None             # kernel us per tick and tick us of the SAME pass (the same script section)
    if len(lus) == n_prof:
        # one launch per tick: price the TIMED ticks' rollback lengths with the per-length kernel means of the instrumented ticks (the script draws a length per
        # tick; 200 other ticks have another mix, which read as a floor fraction above 1 in profiles/r05g)
        by_r = {}
        for r, us in zip(rl[warm + K:warm + K + n_prof], lus): by_r.setdefault(int(r), []).append(us)
        mean_r = {r: sum(v) / len(v) for r, v in by_r.items()}
        k_us = sum(mean_r.get(int(r), k_us) for r in rl[warm:warm + K]) / K
    live = w.active_count(); w.close()
    advances = int(sum(int(r) + 1 for r in rl[warm:warm + K]))
    t = sorted(tick_us)
    out = {"host_loop": "C (benches/tick_loop.c through the C ABI)", "ticks_in_flight": 1, "steps": K, "ms_per_step": secs.value / K * 1e3, "value": live * advances / secs.value, "unit": "entity-frames/s",
           "tick_wall_us": {"median": round(t[K // 2], 2), "p10": round(t[K // 10], 2), "p90": round(t[(9 * K) // 10], 2)},
           "kernel_us": {"mean_under_this_loop": round(k_us, 2), "priced": "per rollback length, weighted by the timed ticks' lengths", "launches_per_tick": round(lpt, 3), "under_the_python_loop": kernel_us}}
    if k_us: out["latency_floor"] = latency_floor(k_us, lpt, secs.value / K * 1e6, same_pass=same, kernel_min_us=min(lus) if lus else None)
    out["platform_floor"] = platform_loop_floor(1, secs.value / K * 1e6, k_us)
    # every Save of every tick (warm-up included) against the oracle under the same script
    from oracle.binding import FLAT, OracleWorld, lib as olib
    olib.gor_set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    o = OracleWorld(n, R + 1, FLAT)
    oids = cm.build_particles(o); cm.spawn_particles(o, oids, n, vel, ttl); o.set_depth(R)
    ok, checked = True, 0
    for f in range(total):
        r = int(rl[f])
        reqs = ([bg.LoadGameState(f - r)] + [x for i in range(r) for x in (([bg.SaveGameState(f - r + i)] if i else []) + [bg.AdvanceFrame((0,))])]) if r else []
        reqs += [bg.SaveGameState(f), bg.AdvanceFrame((0,))]
        if f - R >= 0: o.set_confirmed(f - R)
        want = o.handle_requests(reqs)
        got = [int(cs[2 * R * f + 2 * i]) | (int(cs[2 * R * f + 2 * i + 1]) << 64) for i in range(int(ncs[f]))]
        ok &= got == want; checked += len(want)
    olib.gor_set_num_threads(1)
    out["parity"] = {"checked_ticks": total, "checked_saves": checked, "equal": bool(ok)}
    return out


def single_line(bg, cm, torch, args, dev):
    """One N = 1 SyncTest measurement (BASELINE configs 2 / 3 and the --schema / --entities variants) -> (JSON line, parity failed?)."""
    n, D, K, W = args.entities, args.depth, args.steps, args.warmup
    bps = cm.schema_bytes_per_entity(args.schema)
    m = measure_single(bg, cm, torch, args, contig=False)
    secs, live, gpu_cs, prof = m["secs"], m["live"], m["gpu_cs"], m["prof"]
    # SyncTest's own check over ALL timed ticks (ggrs SyncTestSession: a resimulated frame's checksum must equal the first
    # one recorded for that frame, else MismatchedChecksum): timed tick k at frame F saved frames F-D+1 .. F
    first_seen, resim_ok, f_end = {}, True, m["f_end"]
    for k, tick in enumerate(gpu_cs):
        F = f_end - (len(gpu_cs) - k)
        for j, c in enumerate(tick):
            resim_ok &= first_seen.setdefault(F - D + 1 + j, c) == c
    info = m["info"]
    value = live * (D + 1) * K / secs
    save_ms, save_n = prof["save"]; adv_ms, adv_n = prof["advance"]; load_ms, load_n = prof["load"]; tick_ms, tick_n = prof["tick"]; fin_ms, fin_n = prof["checksum"]
    per = lambda ms, cnt: ms / max(cnt, 1) * 1e-3
    # PMC-derived HBM bytes per launch: only valid for the exact workload the counters were collected on
    traffic = traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    grouped = tick_n > 0
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if "k_tick_hbm_bytes_per_launch" not in tj: tj = tj.get(args.schema, {})     # one entry per schema (scripts/pmc_summary.py)
            if tj.get("entities") == n and tj.get("depth") == D and not args.no_checksum and tj.get("schema", "headline") == args.schema:
                traffic = tj.get("k_tick_hbm_bytes_per_launch" if grouped else "k_copy_state_hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"{tj.get('source', 'profiles/roofline_traffic.json')} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the builder's box; NOT measured in this run)"
        except Exception:
            traffic = None
    if getattr(args, "measure_traffic", False) and grouped:
        # the driver's own record carries MEASURED traffic: two rocprofv3 counter passes of this command's short form, started here after the clock stopped
        live_t = measure_traffic_now(args)
        if live_t.get("hbm_bytes_per_launch"):
            traffic, traffic_source = live_t["hbm_bytes_per_launch"], live_t["source"]
        m["traffic_passes"] = live_t
    save_bytes, tick_bytes = 2 * bps, 2 * bps + 2 * bps * D + ADV_BYTES * (D + 1)
    launches_per_step = 1.0
    if grouped:
        launches_per_step = tick_n / max(min(K, 50), 1)
        full_copy_bytes = bps * (1 + D + 1) * live
        bytes_per_launch = m["prof_bytes"]["tick"] / max(tick_n, 1) if m.get("prof_bytes") else full_copy_bytes
        avg_s = per(tick_ms, tick_n)
        achieved = bytes_per_launch / avg_s / 1e9 if tick_n else 0.0
        kname = info.get("request_group_kernel", "?")
        roof = {"bound": "hbm", "kernel": kname + " -- fused request group: LoadWorld + D x SaveWorld incl. checksums + (D+1) x AdvanceWorld in one launch; checksum rows: " + str(info.get("checksum_fold")),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "frac_compulsory": achieved / HBM_PEAK_GBS,
                "frac_per_request": (tick_bytes * live / avg_s / 1e9 / HBM_PEAK_GBS) if tick_n else 0.0,
                "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_entity": bytes_per_launch / max(live, 1),
                "full_copy_bytes_per_launch": full_copy_bytes, "row_versions": info.get("row_versions"),
                "algorithmic_bytes_note": f"rows the fused launch loads and stores x slots, counted by the library per launch: with row versions only columns whose bytes "
                                          f"differ from the destination's move (every snapshot is complete; the never-written rotation / scale rows are already in every ring slot; HBM traffic can be BELOW this figure: the group's first Save is stored through the L2 and the next launch's loads hit there); "
                                          f"a full copy would move {bps} B/entity snapshot read + {bps} B x saves + {bps} B live write = {bps * (D + 2)} B/entity "
                                          f"(SURVEY 8d's one-kernel-per-request model, {tick_bytes} B/entity-tick, is reported as *_per_request)",
                "avg_launch_us": avg_s * 1e6, "launches_timed": tick_n, "launches_per_step": launches_per_step, "launch_us": m.get("launch_us"),
                "kernarg_bytes": int(info.get("kernarg_bytes", 0) or 0),
                "other_kernels": ({"k_gen_finalize": {"avg_launch_us": per(fin_ms, fin_n) * 1e6, "launches_timed": fin_n}} if fin_n else {}),
                "per_request_equiv_GBps": tick_bytes * live * K / secs / 1e9, "per_request_equiv_frac": tick_bytes * live * K / secs / 1e9 / HBM_PEAK_GBS}
        if tick_n and not args.no_checksum:
            try: roof["alu"] = alu_view(6.0 * 2 * live * D, avg_s, D)
            except Exception: pass
    else:
        save_avg_s = per(save_ms, save_n)
        counted = (m.get("prof_bytes") or {}).get("save", 0)
        save_launch_bytes = counted / max(save_n, 1) if counted else save_bytes * live
        achieved = save_launch_bytes / save_avg_s / 1e9 if save_n else 0.0
        roof = {"bound": "hbm", "kernel": "k_copy_state (SaveWorld)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": save_launch_bytes, "full_copy_bytes_per_launch": save_bytes * live,
                "avg_launch_us": save_avg_s * 1e6, "launches_timed": save_n,
                "other_kernels": {"k_particles_step (AdvanceWorld)": {"avg_launch_us": per(adv_ms, adv_n) * 1e6, "achieved_GBps": ADV_BYTES * live / per(adv_ms, adv_n) / 1e9 if adv_n else 0.0},
                                  "k_copy_state (LoadWorld)": {"avg_launch_us": per(load_ms, load_n) * 1e6, "achieved_GBps": save_bytes * live / per(load_ms, load_n) / 1e9 if load_n else 0.0}},
                "whole_tick_achieved_GBps": tick_bytes * live * K / secs / 1e9, "whole_tick_frac": tick_bytes * live * K / secs / 1e9 / HBM_PEAK_GBS}
    headline = n == 1_000_000 and D == 8 and args.schema == "headline"
    line = {
        "metric": "rollback-resim entity-frames/sec at 1M entities, depth 8; HBM GB/s vs peak" if headline else
                  f"rollback-resim entity-frames/sec at {n} entities, depth {D}, schema {args.schema}; HBM GB/s vs peak",
        "value": value, "unit": "entity-frames/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": secs / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+u64", "data": "synthetic",
        "config": {"workload": f"stress_test {n} entities x {cm.schema_description(args.schema)}, SyncTest depth {D}: 1 load + {D} saves + {D + 1} advances per step",
                   "entities_per_gpu": live, "depth": D, "spawn_system": False, "parallelism": "single GPU",
                   "kernels": "unfused" if args.unfused else ("per-request" if args.no_groups else "request-group"),
                   "arena_actual": info.get("arena"), "request_group_kernel": info.get("request_group_kernel"), "specialised_kernel": info.get("specialised_kernel"),
                   "generated_kernel_origin": info.get("generated_kernel_origin"), "checksum_fold": info.get("checksum_fold"),
                   "hiprtc": info.get("hiprtc"), "device": dev, "nt_stores": bool(args.nt),
                   "host_api": "synchronous handle_requests" if args.sync else "enqueue/collect, 1 tick in flight", **({"DIAGNOSTIC_no_component_checksums": True} if args.no_checksum else {})},
        "preheat": m.get("preheat"), "roofline": roof,
    }
    if m.get("traffic_passes"): line["roofline"]["traffic_passes"] = m["traffic_passes"]
    if grouped and n * bps * (D + 1) <= (256 << 20):
        # the whole ring fits the 256 MB Infinity Cache: launch / latency bound ("report it but do not judge it against HBM peak")
        n_prof_ = max(min(K, 50), 1)
        line["latency_floor"] = latency_floor(per(tick_ms, tick_n) * 1e6, launches_per_step, secs / K * 1e6, same_pass=(tick_ms * 1e3 / n_prof_, m["prof_pass_us"]) if m.get("prof_pass_us") else None,
                                              kernel_min_us=((roof.get("launch_us") or {}).get("min") if launches_per_step == 1.0 else None))
    line["telemetry"] = {"clocks_start": m.get("clocks_start"), "clocks_end": m.get("clocks_end"), "tick_wall_us": m.get("tick_wall_us"), "host_timeline_us_per_tick": m.get("host_timeline"), "rss_mb": m.get("rss_mb")}
    line["parity"] = {"synctest_resim_consistent_over_timed_ticks": bool(resim_ok), "timed_ticks": len(gpu_cs)}
    parity_failed = not resim_ok
    if not args.no_cpu_baseline:
        base, par = cpu_baseline_and_parity(n, D, args.cpu_ticks, m["frames_before_timed"], gpu_cs, 0 if args.no_checksum else args.parity_ticks, schema=args.schema, time_refshaped=not getattr(args, "parity_only", False))
        if base is not None: line["cpu_baseline"] = base
        line["parity"].update(par)
        parity_failed |= par["equal"] is False
    else:
        line["cpu_baseline"] = None
    return line, parity_failed


def p2p_line(bg, cm, torch, args):
    n, D, K, W = args.entities, args.depth, args.steps, args.warmup
    m4 = measure_p2p(bg, cm, torch, args)
    t_ms, t_n = m4["prof"]["tick"]
    avg_s = t_ms / max(t_n, 1) * 1e-3
    bpl = m4["prof_bytes"]["tick"] / max(t_n, 1)
    line = {"metric": f"rollback-resim entity-frames/sec, P2P-shaped rollbacks (0..{D} frames per tick) at {n} entities; GB/s vs HBM peak",
            "value": m4["live"] * m4["advances"] / m4["secs"], "unit": "entity-frames/s", "n_gpus": 1, "steps": K, "warmup": W,
            "ms_per_step": m4["secs"] / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+u64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: p2p-shaped session, {n} entities x 3 registered components, a rollback of 0..{D - 1} frames every tick "
                                   f"(mean {m4['mean_rollback']:.2f}), ring depth {D}", "request_group_kernel": m4["info"].get("request_group_kernel"),
                       "specialised_kernel": m4["info"].get("specialised_kernel"), "specialise_settle": m4["settle"], "checksum_fold": m4["info"].get("checksum_fold"),
                       "arena_actual": m4["info"].get("arena"), "host_api": "enqueue/collect, 1 tick in flight"},
            "roofline": {"bound": "hbm", "kernel": m4["info"].get("request_group_kernel"), "achieved": bpl / avg_s / 1e9 if t_n else 0.0, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bpl / avg_s / 1e9 / HBM_PEAK_GBS if t_n else 0.0, "traffic": None,
                         "avg_launch_us": avg_s * 1e6, "launches_timed": t_n, "algorithmic_bytes_per_launch": bpl,
                         "note": f"the whole ring ({n} x 60 B x {D + 1} blocks = {n * 60 * (D + 1) / 1e6:.0f} MB) lives in the 256 MB Infinity Cache: this line is launch / latency bound, "
                                 "the HBM fraction is reported for completeness, not as its roofline"},
            "latency_floor": latency_floor(avg_s * 1e6, t_n / max(min(K, 50), 1), m4["secs"] / K * 1e6, same_pass=(t_ms * 1e3 / max(min(K, 50), 1), m4["prof_pass_us"])),
            "telemetry": {"tick_wall_us": m4["tick_wall_us"]}, "parity": m4["parity"], "cpu_baseline": m4["cpu_baseline"]}
    return line, not m4["parity"]["equal"]


def fanout_line(bg, cm, torch, args, dist, rank, world_size, dev, ctl_dev):
    """The N > 1 / --fanout / config 5 measurement: speculative fan-out, branches sharded over the ranks (collectives inside libggrs_hip.so)."""
    from bevy_ggrs_amd.fanout import RcclFanout, SpeculativeFanout
    n, D, K, W = args.entities, args.depth, args.steps, args.warmup
    stream = torch.cuda.current_stream().cuda_stream
    flags = (bg.GGRS_WORLD_UNFUSED if args.unfused else 0) | (bg.GGRS_WORLD_NT_COPY if args.nt else 0) | (bg.GGRS_WORLD_NO_GROUPS if args.no_groups else 0)
    # every rank provisions the same world shape; only rank 0 owns the confirmed world, the others receive it through
    # ONE ncclBroadcast of the packed state block.  --spawn (config 5 as SURVEY 8d words it): inputs with INPUT_SPAWN set spawn `rate` particles per
    # frame (particles.rs:258-270), so branches whose predicted input byte carries the bit really diverge from the others
    spawn_rate = 100 if args.spawn else 0
    w = bg.World(n + 2 * spawn_rate * (D + 2), max_depth=D + 2, device=dev, stream=stream, flags=flags)
    ids = cm.build_particles(w, with_spawn=bool(spawn_rate))
    if rank == 0:
        vel, ttl = cm.synthetic_particles(n, ttl="throughput")
        cm.spawn_particles(w, ids, n, vel, ttl)
    else:
        w.spawn(0, {})                                   # seals the world (layout fixed)
    box = [RcclFanout.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    native = RcclFanout(w, rank, world_size, box[0])
    c_rank, comm_size, c_dev = native.comm_info()        # what the communicator says, not what the environment says
    assert c_rank == rank and c_dev == dev, (c_rank, rank, c_dev, dev)
    from bevy_ggrs_amd.fanout import default_branch_input
    bi = default_branch_input
    if os.environ.get("BENCH_BRANCH_INPUT_ZERO") == "1":                 # DIAGNOSTIC: no branch ever holds the spawn key (isolates what the spawn system's presence in the kernel costs)
        bi = lambda b, f: 0
        bi.frame_invariant = True
    fan = SpeculativeFanout(w, dist, depth=D, exchange=None, native=native, branches_per_rank=args.branches, max_inflight=2, branch_input=bi,
                            desync_detection_interval=10 if args.branches == 1 else 1,   # the reference stress_test's default (particles.rs:49, README.md:84)
                            share_prefix=not args.no_share_prefix, spawn_fn=cm.frame_spawn_fn(spawn_rate) if spawn_rate else None,
                            retain=getattr(args, "retain", "none"), compact=not getattr(args, "no_compact", False))
    fan.sync_confirmed(0)
    gc.collect(); gc.disable()                           # see measure_single
    for _ in range(max(W, 18)):                          # (the 16th step of a shape starts the build of the kernel specialised for it)
        fan.step_pipelined(want_result=False)
    fan.drain(want_result=False)
    if not args.no_specialise_wait: w.specialise_wait()
    # pre-heat: EVERY rank must run the same number of steps -- the all-gathers pair up by order (round 4 found the bug a clock-based loop makes)
    pre_t0 = time.perf_counter(); pre_n = 0
    if args.preheat_ms > 0:
        calib = 20
        tc = time.perf_counter()
        for _ in range(calib):
            fan.step_pipelined(want_result=False)
        fan.drain(want_result=False)
        tstep = torch.tensor([(time.perf_counter() - tc) / calib], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tstep, op=dist.ReduceOp.MAX)
        pre_n = calib + int(min(200_000, max(0, args.preheat_ms * 1e-3 / max(float(tstep.item()), 1e-6) - calib)))
        for _ in range(pre_n - calib):
            fan.step_pipelined(want_result=False)
        fan.drain(want_result=False)
    m = {"preheat": {"ms": (time.perf_counter() - pre_t0) * 1e3, "ticks": pre_n, "requested_ms": args.preheat_ms, "same_step_count_on_every_rank": True}}
    P_fan = 0 if (args.no_cpu_baseline or args.no_checksum) else max(0, min(K, args.parity_steps if args.parity_steps >= 0 else max(1, 16 // max(1, world_size * args.branches))))
    c_timed = fan.confirmed
    cf = torch.tensor([c_timed, -c_timed], dtype=torch.int64, device=ctl_dev)
    dist.all_reduce(cf, op=dist.ReduceOp.MAX)                # max(C) == -max(-C): every rank enters the timed region at the same confirmed frame
    if int(cf[0].item()) != -int(cf[1].item()):
        print(f"bench.py: rank {rank} is at confirmed frame {c_timed}, another rank at {int(cf[0].item())} / {-int(cf[1].item())}: the ranks ran different step counts", file=sys.stderr)
        sys.exit(3)
    fan.raw, fan.raw_keep = [], (P_fan if rank == 0 else 0)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fan.step_pipelined(want_result=False)            # enqueue step k+1, collect + all-gather step k
    fan.drain(want_result=False)                                          # every one of the K steps is collected inside the timed region
    w.synchronize()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    gc.enable()
    raw = list(fan.raw)
    fan.raw_keep = 0
    t = torch.tensor([secs], dtype=torch.float64, device=ctl_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    secs = float(t.item())
    live = w.active_count()
    cnt = torch.tensor([live * args.branches], dtype=torch.int64, device=ctl_dev)   # every branch resimulates the whole world
    dist.all_reduce(cnt)
    total_entities = int(cnt.item())
    w.profile_enable(True)
    for _ in range(min(K, 20)):
        fan.step(want_result=False)
    prof = w.profile_read()
    prof_bytes = w.profile_bytes()
    w.profile_enable(False)
    info = w.kernel_info()
    # ---- adoption (--retain): the true inputs of the next D - 1 frames turn out to be what branch 0 predicted (all zero, like the confirmed inputs of this
    # bench): its retained state becomes the world on every rank -- a ring-slot swap on rank 0, a re-simulation (or nothing, at world size 1) elsewhere --, and
    # the world's next SaveWorld must give the checksum the oracle computes for that frame by simulating it in a straight line
    adopt = None
    if fan.retain:
        t_a = time.perf_counter()
        c_before = fan.confirmed
        k_adopt = D - 1 if fan.retain == 4 else D
        fan.adopt(0, k_adopt)
        w.synchronize()
        adopt_ms = (time.perf_counter() - t_a) * 1e3
        cs_after = w.save()
        adopt = {"branch": 0, "from_frame": c_before, "frames_ahead": k_adopt, "frame": fan.confirmed, "ms": round(adopt_ms, 3), "checksum_lo": cs_after & 0xFFFFFFFFFFFFFFFF,
                 "world_frame": w.frame, "mode": "ring-slot swap on the owning rank, re-simulation with the confirmed inputs on the others"}
    value = total_entities * (D + 1) * K / secs
    tick_ms, tick_n = prof["tick"]
    avg_s = tick_ms / max(tick_n, 1) * 1e-3
    bytes_per_launch = prof_bytes["tick"] / max(tick_n, 1)
    # what crosses xGMI per step when N > 1: the all-gather of every rank's Checksum(u128)s (+ a 16-byte tag per step); the confirmed snapshot
    # crossed once, at start-up
    saves_per_rank = fan.saves_per_step
    xgmi = {"all_gather_bytes_per_step_per_rank_sent": 16 * saves_per_rank + 16, "all_gather_bytes_per_step_received": (16 * saves_per_rank + 16) * max(comm_size - 1, 0),
            "steps_per_all_gather": fan.interval, "broadcast_once_bytes": w.state_bytes() if comm_size > 1 else 0,
            "note": "expected collective payload, from the request shapes (no link counters are read): per-link bound only at start-up (flat broadcast of the packed state block)"}
    line = {
        "metric": "rollback-resim entity-frames/sec at 1M entities, depth 8; HBM GB/s vs peak" if (n == 1_000_000 and D == 8 and args.branches == 1) else
                  f"rollback-resim entity-frames/sec at {n} entities, depth {D}, {args.branches} predicted-input branches per rank; GB/s vs peak",
        "value": value, "unit": "entity-frames/s", "n_gpus": comm_size, "steps": K, "warmup": W,
        "ms_per_step": secs / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+u64", "data": "synthetic",
        "config": {"workload": f"stress_test {n} entities x {cm.schema_description(args.schema)}, speculative fan-out: per step and branch 1 load + {D} saves + {D + 1} advances",
                   "entities_per_gpu": live, "depth": D, "spawn_system": bool(args.spawn),
                   "parallelism": f"speculative fan-out, {args.branches} predicted-input branch(es) per rank x {comm_size} ranks (ncclCommCount) (ncclBroadcast of the confirmed snapshot once, one ncclAllGather of the checksums per {fan.interval} step(s) on a side stream -- both inside libggrs_hip.so, ggrs_hip_fanout_*)",
                   "kernels": "request-group", "arena_actual": info.get("arena"), "request_group_kernel": info.get("request_group_kernel"), "specialised_kernel": info.get("specialised_kernel"),
                   "hiprtc": info.get("hiprtc"), "device": dev, "host_api": "ggrs_hip_fanout_step, 2 steps in flight"},
        "preheat": m.get("preheat"),
        "roofline": {"bound": "hbm", "kernel": str(info.get("request_group_kernel")) + " (rank 0)", "achieved": bytes_per_launch / avg_s / 1e9 if tick_n else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_per_launch / avg_s / 1e9 / HBM_PEAK_GBS if tick_n else 0.0, "traffic": None, "avg_launch_us": avg_s * 1e6, "launches_timed": tick_n,
                     "launches_per_step": tick_n / max(min(K, 20), 1), "algorithmic_bytes_per_launch": bytes_per_launch},
        "xgmi_expected": xgmi,
    }
    if fan.retain:
        line["config"]["retain"] = {2: "newest", 4: "all"}[fan.retain]
        line["config"]["workload"] += f"; every branch's frames are KEPT in private state blocks ({'all ' + str(D) if fan.retain == 4 else 'the newest'} per branch, row versions apply)"
        line["roofline"]["note"] = ("retained branch states: the launch's algorithmic bytes are the source snapshot read once per branch + every retained frame's hot rows stored "
                                    "(ggrs_hip_profile_read_bytes); next to roofline_alu, which prices the same launch's hashing")
    if args.branches > 1:
        # checksum-only branches (dead-snapshot elimination): integer-multiply bound, not HBM bound.  Roofline = SeaHash `diffuse`
        # per second against the chip's measured ceiling (scripts/ubench_alu.hip -> profiles/alu_ceiling.json).
        try: ceil = json.load(open(os.path.join(ROOT, "profiles", "alu_ceiling.json")))
        except Exception: ceil = None
        diffuses = 6.0 * 2 * live * D * args.branches * comm_size * K
        line["roofline_alu"] = {"bound": "valu-int (u64 multiply)", "achieved": diffuses / secs / 1e9, "unit": "G diffuse/s",
                                "peak": (ceil or {}).get("diffuse_G_per_s"), "frac": (diffuses / secs / 1e9 / ceil["diffuse_G_per_s"]) if ceil else None,
                                "peak_source": (ceil or {}).get("source"), "shared_prefix": not args.no_share_prefix,
                                "note": "algorithmic diffuses: 6 per entity per checksummed component per SaveWorld, 2 components, D SaveWorlds per branch -- every branch's D "
                                        "Checksum(u128)s are delivered.  The kernel hoists the order hash and memoises unchanged tails, and the step computes the "
                                        "branch-invariant Save(C+1) once per rank instead of once per branch (shared_prefix), so fewer are executed"}
        line["roofline_alu"].update(alu_executed(line["roofline_alu"]["frac"], D))
    parity_failed = False
    if rank == 0 and not args.no_cpu_baseline:
        if not getattr(args, "parity_only", False):
            base, _ = cpu_baseline_and_parity(n, D, args.cpu_ticks, D + 1, [], 0)
            line["cpu_baseline"] = base
        from bevy_ggrs_amd.fanout import default_branch_input
        par = fanout_parity(n, D, c_timed, raw, comm_size, args.branches, bi, lambda f: 0,
                            threads=max(1, min(64, os.cpu_count() or 1)), spawn_rate=100 if args.spawn else 0)
        par["cross_rank_confirmed_frames_agree"] = True     # SpeculativeFanout raises DesyncDetected otherwise (every step, every rank)
        if adopt is not None:
            want = oracle_checksum_at(n, D, adopt["frame"], lambda f: 0, spawn_rate=100 if args.spawn else 0)
            adopt["equal_to_oracle_straight_line"] = bool(want == cs_after and adopt["world_frame"] == adopt["frame"])
            adopt["oracle_checksum_lo"] = want & 0xFFFFFFFFFFFFFFFF
            par["adopt"] = adopt
            if not adopt["equal_to_oracle_straight_line"]: par["equal"] = False
        line["parity"] = par
        parity_failed = par["equal"] is not True
    elif rank == 0:
        line["cpu_baseline"] = None
    native.close() if hasattr(native, "close") else None
    w.close()
    return line, parity_failed


def boundary_is_the_number(line):
    """Configs 2 and 4 are bound by the host loop around a 5-10 us kernel: the graded boundary is the C ABI, so the line's `value` / `ms_per_step` / `latency_floor`
    are what a C host gets through it (benches/tick_loop.c, one tick in flight); what bench.py's own ctypes loop gets moves under `telemetry.python_driver`."""
    c = line.get("c_loop") or {}
    if "value" not in c: return
    line.setdefault("telemetry", {})["python_driver"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "steps": line.get("steps"), "latency_floor": line.get("latency_floor"),
                                                         "note": "bench.py's ctypes loop around the same C ABI: marshalling and field writes per tick are test plumbing, not the boundary"}
    line["value"], line["ms_per_step"], line["steps"] = c["value"], c["ms_per_step"], c["steps"]
    if c.get("latency_floor"): line["latency_floor"] = c["latency_floor"]
    line["host_loop"] = c.get("host_loop")


def floors_inconsistent(line):
    """A latency floor above the tick it was priced against means the model's inputs disagree: that fails the line (it used to be capped at 1)."""
    bad = []
    for where, lf in (("latency_floor", line.get("latency_floor")), ("c_loop.latency_floor", (line.get("c_loop") or {}).get("latency_floor")),
                      ("python_driver.latency_floor", ((line.get("telemetry") or {}).get("python_driver") or {}).get("latency_floor"))):
        if lf and lf.get("consistent") is False and lf.get("tick_us_same_pass"): bad.append(where)
    if bad: line["floor_inconsistent"] = bad
    return bool(bad)


def compact(line, keep=("value", "unit", "ms_per_step", "steps", "warmup", "parity", "latency_floor", "roofline_alu", "c_loop", "xgmi_expected", "host_loop", "floor_inconsistent")):
    """An extra_configs entry: the figures and their evidence, without the headline's long notes."""
    out = {k: line[k] for k in keep if k in line}
    out["workload"] = line.get("config", {}).get("workload")
    r = line.get("roofline") or {}
    out["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_timed", "launches_per_step", "algorithmic_bytes_per_launch", "kernarg_bytes") if k in r}
    if "alu" in r: out["roofline"]["alu_frac"] = (r["alu"] or {}).get("frac")
    tw = (line.get("telemetry") or {}).get("tick_wall_us") or {}
    out["tick_wall_us_median"] = tw.get("median")
    ht = (line.get("telemetry") or {}).get("host_timeline_us_per_tick")
    if ht: out["host_timeline_us_per_tick"] = ht
    pd = (line.get("telemetry") or {}).get("python_driver")
    if pd: out["python_driver"] = pd
    return out


def extra_configs(bg, cm, torch, base_args, dev, budget_s=60.0):
    """every other BASELINE config + the all-columns-hot world in the driver's record -- measured in this process AFTER the headline's clock has
    stopped, each with its own in-run oracle parity.  Short forms (fewer steps, 40 ms pre-heat, parity on the first timed ticks only, no CPU timing legs)."""
    import copy
    import torch.distributed as dist
    out, failed = {}, False
    t0 = time.perf_counter()

    def mk(**over):
        a = copy.copy(base_args)
        a.preheat_ms, a.parity_ticks, a.cpu_ticks, a.parity_only, a.schema, a.sync, a.no_cpu_baseline = 40.0, 6, 1, True, "headline", False, False
        for k, v in over.items(): setattr(a, k, v)
        return a

    def guard(name, fn):
        nonlocal failed
        if time.perf_counter() - t0 > budget_s:
            out[name] = {"skipped": f"extra_configs budget of {budget_s:.0f} s spent"}; return
        t1 = time.perf_counter()
        try:
            line, bad = fn()
            c = compact(line); c["seconds"] = round(time.perf_counter() - t1, 1)
            out[name] = c; failed |= bool(bad)
        except Exception as e:              # noqa: BLE001 -- an extra line must not take the headline down; its absence is visible
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ---- config 2: 10 k entities (launch / latency bound) -- bench.py's own loop and the C loop
    def cfg2():
        a = mk(entities=10_000, steps=400, warmup=16)
        line, bad = single_line(bg, cm, torch, a, dev)
        kus = line["roofline"]["avg_launch_us"]
        line["c_loop"] = c_loop_synctest(bg, cm, torch, a.entities, a.depth, 2000, kernel_us=kus)
        line["c_loop"]["two_in_flight"] = {k: v for k, v in c_loop_synctest(bg, cm, torch, a.entities, a.depth, 2000, inflight=2, kernel_us=kus, parity_ticks=0).items() if k in ("ms_per_step", "value", "latency_floor", "ticks_in_flight", "kernel_us")}
        boundary_is_the_number(line)
        return line, bad or line["c_loop"]["parity"]["equal"] is not True or floors_inconsistent(line)
    guard("config2", cfg2)
    # ---- config 4: P2P-shaped rollbacks at 100 k
    def cfg4():
        a = mk(entities=100_000, steps=200, warmup=16, cpu_ticks=0, no_cpu_baseline=True)
        line, bad = p2p_line(bg, cm, torch, a)
        r = line["roofline"]
        line["c_loop"] = c_loop_p2p(bg, cm, torch, a.entities, a.depth, 600, kernel_us=r["avg_launch_us"], launches_per_tick=line["latency_floor"]["launches_per_tick"])
        boundary_is_the_number(line)
        return line, bad or line["c_loop"]["parity"]["equal"] is not True or floors_inconsistent(line)
    guard("config4", cfg4)
    # ---- the all-columns-hot world: every Save moves all 15 rows (what the reference's clone-everything save always does)
    guard("allhot", lambda: single_line(bg, cm, torch, mk(schema="allhot", steps=40, warmup=8), dev))
    # ---- config 5 on this GPU: 256 predicted-input branches x 100 k x 8 frames (world size 1 over the real RCCL), without and with the spawn system
    if dist.is_available():
        own = not dist.is_initialized()
        try:
            if own:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
            guard("config5_1gpu", lambda: fanout_line(bg, cm, torch, mk(entities=100_000, branches=256, steps=10, warmup=2, parity_steps=1, spawn=False, fanout=True), dist, 0, 1, dev, f"cuda:{dev}"))
            guard("config5_retain", lambda: fanout_line(bg, cm, torch, mk(entities=100_000, branches=256, steps=10, warmup=2, parity_steps=1, spawn=False, fanout=True, retain="all"), dist, 0, 1, dev, f"cuda:{dev}"))
            guard("config5_spawn", lambda: fanout_line(bg, cm, torch, mk(entities=100_000, branches=256, steps=6, warmup=2, parity_steps=1, spawn=True, fanout=True, preheat_ms=0.0), dist, 0, 1, dev, f"cuda:{dev}"))
        except Exception as e:              # noqa: BLE001
            out.setdefault("config5_1gpu", {"error": f"{type(e).__name__}: {e}"[:300]})
        finally:
            if own and dist.is_initialized(): dist.destroy_process_group()
    out["seconds_total"] = round(time.perf_counter() - t0, 1)
    return out, failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--entities", type=int, default=1_000_000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ticks", type=int, default=3)
    ap.add_argument("--parity-ticks", type=int, default=24, help="timed ticks whose checksums the CPU oracle replays and compares (N = 1 only; 0 = off)")
    ap.add_argument("--no-specialise-wait", action="store_true", help="do not wait for the kernel specialised for the tick's group shape before the timed region")
    ap.add_argument("--preheat-ms", type=float, default=150.0, help="wall time of untimed ticks between warm-up and the timed region (clock ramp; reported as `preheat`)")
    ap.add_argument("--unfused", action="store_true", help="one kernel per reference system (no fusion at all)")
    ap.add_argument("--no-groups", action="store_true", help="one launch per request (no request-group fusion)")
    ap.add_argument("--nt", action="store_true", help="non-temporal snapshot copies (A/B knob)")
    ap.add_argument("--sync", action="store_true", help="synchronous ggrs_hip_handle_requests per step (host blocks on every tick) "
                    "instead of the default enqueue/collect pipeline (tick N+1 is enqueued before tick N's checksums are collected)")
    ap.add_argument("--schema", choices=["headline", "full", "allhot"], default="headline",
                    help="headline: BASELINE's 3 registered components (60 B/entity); full: the reference stress_test's POD schema "
                         "(+ GlobalTransform 12 x f32, three 1-byte visibilities: examples/stress_tests/particles.rs:190-199); allhot: the headline "
                         "components with every column written every frame (+ increase_component over rotation / scale): every Save moves all 15 rows")
    ap.add_argument("--fanout", action="store_true", help="run the N > 1 code path (RCCL broadcast + all-gather inside the library) even at world size 1")
    ap.add_argument("--branches", type=int, default=1, help="fan-out path: predicted-input branches per rank (BASELINE config 5: 256 over all ranks)")
    ap.add_argument("--spawn", action="store_true", help="fan-out: register the stress_test's spawn system (100 particles per frame while INPUT_SPAWN is held): the "
                    "branches whose predicted input byte carries the bit diverge from the others (SURVEY 8d's wording of config 5).  The spawn runs INSIDE the branch's "
                    "request group (fused), and identical spawning branches -- same frames, same staged payload -- still ride in one launch")
    ap.add_argument("--retain", choices=["none", "newest", "all"], default="none", help="fan-out: keep the branches' frames in private state blocks (GGRS_BRANCH_RETAIN_*) so that a "
                    "matching branch can be ADOPTED when the true input arrives (SURVEY 8e); after the timed region the bench adopts one and checks the adopted world against the oracle")
    ap.add_argument("--no-compact", action="store_true", help="fan-out A/B: hand the library the step as a request list (rounds 3-5) instead of ggrs_hip_fanout_step_branches")
    ap.add_argument("--no-share-prefix", action="store_true", help="fan-out A/B: every branch replays [Load(C), Advance(confirmed input), Save(C+1)] itself "
                    "(the round-3 request lists) instead of starting from the ONE saved C+1")
    ap.add_argument("--parity-steps", type=int, default=-1, help="N > 1 / --fanout: timed steps whose gathered checksum table rank 0 replays on the CPU oracle "
                    "(every branch of every rank); default: about 16 branch walks in all, at least one step")
    ap.add_argument("--control-backend", choices=["nccl", "gloo"], default=None, help="torch.distributed backend of the control plane (unique id, barrier, max over "
                    "ranks); default nccl, gloo when ranks share a device.  The data-path collectives are always issued inside libggrs_hip.so")
    ap.add_argument("--oversubscribe", action="store_true", help="--gpus N with fewer than N visible devices: rank r runs on device r %% devices (correctness only)")
    ap.add_argument("--dry-run", action="store_true", help="--gpus N: print the N rank command lines (JSON, one per line) and exit")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=3,
                    help="BASELINE.json config: 3 (default) = the headline, stress_test 1 M x depth 8; 2 = 10 k entities; 4 = P2P-shaped rollbacks "
                         "(0..8 frames per tick) at 100 k; 5 = 256 predicted-input branches x 100 k x 8 frames (all on this node's GPUs)")
    ap.add_argument("--no-extra", action="store_true", help="headline only: skip `extra_configs` (configs 2 / 4 / 5, the all-columns-hot world), which the default N = 1 headline run "
                    "measures after its clock has stopped")
    ap.add_argument("--extra-budget-s", type=float, default=150.0, help="wall-time budget of `extra_configs`: configs that would start beyond it are reported as skipped")
    ap.add_argument("--no-traffic", action="store_true", help="headline: do not start the two rocprofv3 counter passes that put MEASURED HBM bytes per launch into roofline.traffic")
    ap.add_argument("--no-lazy-live", action="store_true", help="A/B: every tick writes the live block (the library's test hook ggrs_dbg_set_lazy_live)")
    ap.add_argument("--no-checksum", action="store_true", help="DIAGNOSTIC ONLY: no component checksums registered (isolates the hash ALU cost; not a valid bench line)")
    args = ap.parse_args()
    if args.no_lazy_live: os.environ["BENCH_NO_LAZY_LIVE"] = "1"
    default_headline = (args.config == 3 and args.entities == 1_000_000 and args.depth == 8 and args.schema == "headline" and not (args.fanout or args.sync or args.unfused or args.no_groups
                        or args.nt or args.no_checksum or args.no_cpu_baseline or args.branches != 1))
    args.measure_traffic = default_headline and not args.no_traffic and args.gpus == 1
    if args.config == 2: args.entities = 10_000
    if args.config == 4: args.entities = 100_000
    if args.config == 5:
        args.entities, args.fanout = 100_000, True
        if args.branches == 1: args.branches = max(1, 256 // max(1, args.gpus))

    # ---- `--gpus N` with no launcher around this process: start the N ranks here
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, [a for a in sys.argv[1:] if a != "--dry-run"]))
    if args.dry_run:
        print(json.dumps({"env": {}, "cmd": [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]}))
        return

    # stdout carries ONE JSON line and nothing else.  Native libraries write there too -- RCCL prints a version banner through C stdio, which leaves its buffer at
    # exit, i.e. AFTER the line (seen in profiles/r06o): from here on file descriptor 1 is stderr, and the line goes to the descriptor stdout was.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import __graft_entry__ as ge
    ge.build()
    import bevy_ggrs_amd as bg
    import common as cm

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size and "WORLD_SIZE" in os.environ:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world_size} rank(s)", file=sys.stderr); sys.exit(2)
    dist = None
    distributed = world_size > 1 or args.fanout
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py: no GPU visible (the product path has no CPU fallback)", file=sys.stderr); sys.exit(2)
    if local_rank >= ndev and not args.oversubscribe:
        print(f"bench.py: rank {rank} needs device {local_rank} but only {ndev} visible (--oversubscribe shares devices)", file=sys.stderr); sys.exit(2)
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    ctl_dev = f"cuda:{dev}"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        # torch.distributed is the CONTROL plane only (the ncclUniqueId, the timing barrier, the max over ranks); the data-path
        # collectives are issued inside libggrs_hip.so.  Ranks that share a device (--oversubscribe: correctness runs on a one-GPU
        # box) cannot form a torch NCCL group -- RCCL refuses two ranks per device -- so their control plane is gloo.
        backend = args.control_backend or ("gloo" if world_size > ndev else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
            ctl_dev = "cpu"

    if distributed:
        line, parity_failed = fanout_line(bg, cm, torch, args, dist, rank, world_size, dev, ctl_dev)
        dist.barrier()
        dist.destroy_process_group()
    elif args.config == 4:
        line, parity_failed = p2p_line(bg, cm, torch, args)
    else:
        line, parity_failed = single_line(bg, cm, torch, args, dev)
        if default_headline and not args.no_extra:
            # every other BASELINE config, measured AFTER the headline's timed region (and its parity / CPU legs) in the same process
            extra, extra_failed = extra_configs(bg, cm, torch, args, dev, budget_s=args.extra_budget_s)
            line["extra_configs"] = extra
            parity_failed |= extra_failed
    if rank == 0:
        os.write(line_fd, (json.dumps(line) + "\n").encode())
        if parity_failed:
            print("bench.py: PARITY FAILURE -- GPU checksums differ from the CPU oracle's (see \"parity\" of the line / of its extra_configs)", file=sys.stderr)
            sys.exit(1)


if __name__ == "__main__":
    main()
