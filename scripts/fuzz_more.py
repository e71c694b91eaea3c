#!/usr/bin/env python
"""One-off wider fuzz (not part of the suite): the branch-step / adoption fuzz and the device-spawn fuzz of tests/ over seed ranges given on the command line.
usage: fuzz_more.py branches 8000 8120 | devspawn 9100 9140      prints one JSON line per failure and a summary"""
import json, multiprocessing as mp, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bad = []
    if kind == "branches":
        import test_gpu_zfuzz_branches as t
        ctx = mp.get_context("spawn")
        for s0 in range(lo, hi, 12):
            seeds = list(range(s0, min(hi, s0 + 12)))
            q = ctx.Queue(); p = ctx.Process(target=t._fuzz_rank, args=(q, seeds)); p.start()
            try: r = q.get(timeout=900)
            except Exception as e: r = ("error", f"no answer: {e}", "")
            p.join(timeout=30)
            if p.is_alive(): p.kill()
            if r[0] != "ok": bad.append({"seeds": seeds, "error": str(r[1])[:600]}); print(json.dumps(bad[-1]), flush=True)
    else:
        import test_gpu_device_spawn as t
        for seed in range(lo, hi):
            try: t.test_device_spawns_fuzzed(seed)
            except Exception as e:                             # noqa: BLE001
                bad.append({"seed": seed, "error": f"{type(e).__name__}: {e}"[:600], "tb": traceback.format_exc()[-600:]}); print(json.dumps(bad[-1]), flush=True)
    print(json.dumps({"kind": kind, "seeds": [lo, hi], "failures": len(bad)}))


if __name__ == "__main__":
    main()
