#!/usr/bin/env python
"""rocprofv3 kernel trace -> steady-state launch duration of the dominant kernel.

`rocprofv3 --stats` averages over ALL calls of a kernel; bench.py's run holds 9 short ring warm-up groups
([Save, Advance] before the first LoadGameState) next to the tick-shaped launches the roofline is quoted on.
This splits them (threshold: half the longest launch) so the figure can be compared with roofline.avg_launch_us.

usage: kernel_trace_steady.py <dir with *kernel_trace.csv> <out.json>
"""
import csv
import glob
import json
import os
import statistics
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    out = {"source": files}
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, tv in per.items():
        if not ("ggrs::k_tick" in name or "ggrs_jit_tick" in name or "k_gen_finalize" in name):
            continue
        v = [d for _, d in sorted(tv)]
        big = [x for x in v if x > max(v) / 2]
        # world creation times a few tick-shaped launches on every candidate arena (placement probe);
        # the bench's own ticks are the LAST ones of the trace
        last = big[-100:]
        out[name] = {"all_calls": {"n": len(v), "mean_us": statistics.mean(v) / 1e3},
                     "tick_shaped_launches": {"n": len(big), "mean_us": statistics.mean(big) / 1e3,
                                              "min_us": min(big) / 1e3, "max_us": max(big) / 1e3},
                     "last_100_tick_shaped_launches": {"n": len(last), "mean_us": statistics.mean(last) / 1e3,
                                                       "min_us": min(last) / 1e3, "max_us": max(last) / 1e3},
                     # the default bench command: 150 ms of pre-heat ticks, 16 warm-up, 200 timed -- the timed region is the last 200
                     "last_200_tick_shaped_launches": {"n": len(big[-200:]), "mean_us": statistics.mean(big[-200:]) / 1e3,
                                                       "median_us": statistics.median(big[-200:]) / 1e3,
                                                       "min_us": min(big[-200:]) / 1e3, "max_us": max(big[-200:]) / 1e3}}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
