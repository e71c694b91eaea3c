#!/usr/bin/env python
"""The reference's own criterion benchmarks (benches/bench.rs), re-expressed over the C ABI.

  advance_and_load_1000_components / advance_and_save_1000_components            (bench.rs:48-66)
  advance_and_load_3000_disjoint_components / advance_and_save_3000_disjoint_components   (bench.rs:68-95)

Each benchmark function is ONE iteration of criterion's loop:
  advance_and_load:  run AdvanceWorld; RollbackFrameCount = 0; run LoadWorld        (bench.rs:19-23)
  advance_and_save:  run AdvanceWorld; run SaveWorld                                 (bench.rs:25-28)
These worlds are 4-12 KB: the GPU numbers are pure launch latency (2 kernel launches per iteration) and are reported for
completeness, next to the same functions on the CPU oracle's reference-shaped storage (1 thread).  The reference ships
no results for them.  Not the headline metric: see bench.py.

usage: python benches/criterion_mirror.py [--iters N] [--scale K]   (K multiplies the 1000 entities per component)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def build(world, bg, disjoint, n):
    foo = world.register_component("Foo", 4, 1)
    comps = [foo]
    world.add_system(bg.SYS_ADD_U32, comp=(foo,), word=(0,), iparam=(1,))                      # increment_foos
    if disjoint:
        bar = world.register_component("Bar", 4, 1); baz = world.register_component("Baz", 4, 1)
        world.add_system(bg.SYS_ADD_U32, comp=(bar,), word=(0,), iparam=(-1 & 0xFFFFFFFF,))    # decrement_bars
        world.add_system(bg.SYS_ADD_U32, comp=(baz,), word=(0,), iparam=(1,))                  # increment_bazs
        comps += [bar, baz]
    v = np.arange(n, dtype=np.uint32)
    for c in comps:
        world.spawn(n, {c: [v]})                   # (bench.rs spawns Foo(i), Bar(i), Baz(i) interleaved; sets are disjoint either way)
    world.set_depth(8)
    world.save()                                   # app.world_mut().run_schedule(SaveWorld)
    return comps


def advance_and_load(w):
    w.advance()
    w.set_frame(0)
    w.load(0)


def advance_and_save(w):
    w.advance()
    w.save()


def time_fn(w, fn, iters, sync):
    for _ in range(20):
        fn(w)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn(w)
    sync()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--scale", type=int, default=1)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import bevy_ggrs_amd as bg
    from oracle.binding import REFSHAPED, OracleWorld, lib
    lib.gor_set_num_threads(1)
    n = 1000 * args.scale
    out = {"benchmarks": {}, "entities_per_component": n, "iters": args.iters,
           "note": "one iteration = the body of the reference's criterion loop (benches/bench.rs:19-28); microseconds per iteration"}
    for disjoint, tag in ((False, f"{n}_components"), (True, f"{3 * n}_disjoint_components")):
        for fn, name in ((advance_and_load, "advance_and_load_"), (advance_and_save, "advance_and_save_")):
            g = bg.World(3 * n + 8, max_depth=8)
            build(g, bg, disjoint, n)
            o = OracleWorld(3 * n + 8, 8, REFSHAPED)
            build(o, bg, disjoint, n)
            out["benchmarks"][name + tag] = {
                "gpu_us": time_fn(g, fn, args.iters, g.synchronize),
                "cpu_reference_shaped_1_thread_us": time_fn(o, fn, max(50, args.iters // 10), lambda: None),
            }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
