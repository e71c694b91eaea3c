#!/usr/bin/env python
"""rocprofv3 counter CSVs -> profiles/<tag>/pmc_summary.json (+ profiles/roofline_traffic.json).

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE `rocprofv3 --pmc` passes (they do not fit one
pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units are KiB per dispatch.
Corrections, as that guide prescribes: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by
exactly 2x on gfx950 -> doubled; WRITE_SIZE is calibrated against this engine's own known byte
count (one snapshot copy writes 60.5 MB -> reported 59 082 KiB: exact, no correction).

usage: pmc_summary.py <gpurun_out/tag> <profiles/tag> [schema [prefix]]
       schema: which bench.py --schema the profiled command ran (default headline); prefix: the pass directories are <prefix>_fetch / <prefix>_write
       (default prof).  profiles/roofline_traffic.json keeps one entry per schema.
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "ggrs" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    schema = sys.argv[3] if len(sys.argv) > 3 else "headline"
    prefix = sys.argv[4] if len(sys.argv) > 4 else "prof"
    os.makedirs(dst, exist_ok=True)
    out = {"note": __doc__.split("usage:")[0].strip()}
    fetch = collect(os.path.join(src, prefix + "_fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = collect(os.path.join(src, prefix + "_write", "**", "*counter_collection.csv"), "WRITE_SIZE")
    for k in sorted(set(fetch) | set(write)):
        d = {}
        for name, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            v = agg.get(k, [])
            v = v[len(v) // 2:]                       # steady state: second half of the dispatches
            if v:
                d[name + "_KiB_mean"] = sum(v) / len(v)
                d["dispatches_" + name] = len(agg[k])
        if "FETCH_SIZE_KiB_mean" in d and "WRITE_SIZE_KiB_mean" in d:
            # the x2 FETCH correction is the guide's calibration for WIDE (16 B per lane) coalesced reads: k_copy_state (and round 3's k_tick3).  The
            # generated kernel reads 4 bytes per lane (uncalibrated width): its FETCH_SIZE is reported as is AND doubled, and only
            # WRITE_SIZE (calibrated against a known snapshot copy) is relied on -- reads are 10 % of its traffic.
            wide = "ggrs_jit" not in k
            d["hbm_bytes_per_launch_corrected"] = ((2 if wide else 1) * d["FETCH_SIZE_KiB_mean"] + d["WRITE_SIZE_KiB_mean"]) * 1024
            d["fetch_correction"] = "x2 (16 B per lane reads, MI355X_MICROARCH.md)" if wide else "none (4 B per lane reads: uncalibrated width; with x2 the total would be %.0f bytes)" % ((2 * d["FETCH_SIZE_KiB_mean"] + d["WRITE_SIZE_KiB_mean"]) * 1024)
        out[k] = d
    summary_name = "pmc_summary.json" if schema == "headline" else f"pmc_summary_{schema}.json"
    json.dump(out, open(os.path.join(dst, summary_name), "w"), indent=1)
    roof = {}
    for k, d in out.items():
        if isinstance(d, dict) and "hbm_bytes_per_launch_corrected" in d:
            if "ggrs_jit_tick" in k and d.get("dispatches_FETCH_SIZE", 0) >= 20 and d.get("dispatches_FETCH_SIZE", 0) >= roof.get("_n", 0):
                roof["_n"] = d["dispatches_FETCH_SIZE"]                # the kernel that served the bench's ticks: the one with the most dispatches
                roof["k_tick_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
                roof["kernel"] = k
            if "k_copy_state" in k:
                roof["k_copy_state_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
    roof.pop("_n", None)
    if roof:
        roof["source"] = os.path.join(dst, summary_name)
        roof["entities"] = 1000000; roof["depth"] = 8; roof["schema"] = schema       # the workload scripts/gpu_round4.sh profiles (bench.py defaults)
        tpath = os.path.join(os.path.dirname(dst.rstrip("/")), "roofline_traffic.json")
        try:
            cur = json.load(open(tpath))
        except Exception:
            cur = {}
        if "k_tick_hbm_bytes_per_launch" in cur:                                        # the round-3 flat form: one entry, the headline schema
            cur = {cur.get("schema", "headline"): cur}
        cur[schema] = roof
        json.dump(cur, open(tpath, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
