#!/usr/bin/env python
"""Where does the FIRST timed tick's time go (86-108 us against a 45 us kernel)?  Reads a `rocprofv3 --hip-trace --kernel-trace -f csv`
output directory of `bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline` and lines the first timed ticks up: when the host called the launch, when
the call returned, when the kernel began and ended on the device, and what ran just before.  usage: first_tick_trace.py <dir> [out.json]"""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    if not f: return []
    return list(csv.DictReader(open(f[0])))


def main():
    d = sys.argv[1]
    k = rows(os.path.join(d, "**", "*kernel_trace.csv"))
    h = rows(os.path.join(d, "**", "*hip_api_trace.csv"))
    ticks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Correlation_Id")) for r in k if r["Kernel_Name"].startswith("ggrs_jit_tick")), key=lambda x: x[0])
    allk = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k), key=lambda x: x[0])
    # the timed region: the LAST idle gap of > 40 us that is followed by at least 20 tick kernels with gaps < 40 us between them (bench.py synchronises the device before it starts the clock)
    start = None
    for i in range(len(ticks) - 20, 0, -1):
        if ticks[i][0] - ticks[i - 1][1] > 40_000 and all(ticks[j + 1][0] - ticks[j][1] < 40_000 for j in range(i, i + 19)):
            start = i; break
    if start is None:
        print(json.dumps({"error": "no timed region found", "tick_kernels": len(ticks)})); return
    launches = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Correlation_Id")) for r in h if "LaunchKernel" in r["Function"]), key=lambda x: x[0])
    by_corr = {c: (s, e, f) for s, e, f, c in launches}
    out = {"timed_region_first_tick_kernel_index": start, "ticks": []}
    t0 = None
    for n, i in enumerate(range(start, start + 6)):
        ks, ke, corr = ticks[i]
        api = by_corr.get(corr)
        prev_end = ticks[i - 1][1]
        before = [x for x in allk if x[1] <= ks and x[1] > ks - 200_000][-3:]
        if t0 is None: t0 = api[0] if api else ks
        out["ticks"].append({"tick": n, "launch_call_start_us": round(((api[0] if api else ks) - t0) / 1e3, 2), "launch_call_us": round((api[1] - api[0]) / 1e3, 2) if api else None,
                             "call_start_to_kernel_begin_us": round((ks - api[0]) / 1e3, 2) if api else None, "kernel_us": round((ke - ks) / 1e3, 2),
                             "kernel_begin_us": round((ks - t0) / 1e3, 2), "kernel_end_us": round((ke - t0) / 1e3, 2), "gap_after_previous_tick_kernel_us": round((ks - prev_end) / 1e3, 2),
                             "kernels_just_before": [(x[2][:24], round((ks - x[1]) / 1e3, 1)) for x in before]})
    steady = [ticks[i][1] - ticks[i][0] for i in range(start + 5, start + 20)]
    out["steady_kernel_us_median"] = round(sorted(steady)[len(steady) // 2] / 1e3, 2)
    out["steady_kernel_to_kernel_gap_us_median"] = round(sorted(ticks[i + 1][0] - ticks[i][1] for i in range(start + 5, start + 19))[7] / 1e3, 2)
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 2: open(sys.argv[2], "w").write(s)


if __name__ == "__main__":
    main()
