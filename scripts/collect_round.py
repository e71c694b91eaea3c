#!/usr/bin/env python
"""gpurun_out/<tag> (scratch) -> profiles/<tag> (committed): the small files of scripts/gpu_round2.sh plus the rocprofv3
summaries of the two profiled commands (bench.py = k_tick3; tick_bench under GGRS_TICK_GENERIC=1 = the generated kernel).

usage: collect_round.py <tag>"""
import collections, csv, glob, json, os, shutil, subprocess, sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "*")):
    if os.path.isfile(f) and os.path.getsize(f) < 2 << 20 and not f.endswith((".err", ".log")) or f.endswith(("pytest_gpu.log", "smoke.log")):
        shutil.copy(f, dst)
for name, sub in (("kernel_stats.csv", "prof_stats"), ("jit_kernel_stats.csv", "prof_jit_stats")):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits: shutil.copy(hits[0], os.path.join(dst, name))
subprocess.check_call([sys.executable, "scripts/pmc_summary.py", src, dst], stdout=subprocess.DEVNULL)


def collect(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter: agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


fe = collect(os.path.join(src, "prof_jit_fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
wr = collect(os.path.join(src, "prof_jit_write", "**", "*counter_collection.csv"), "WRITE_SIZE")
out = {"note": "GGRS_TICK_GENERIC=1 benches/tick_bench 1000000 8 100 16 0 0 1 (stress_test world served by the generated kernel); FETCH_SIZE / "
               "WRITE_SIZE in separate rocprofv3 --pmc passes, KiB per dispatch, mean over the second half of the dispatches; FETCH doubled "
               "(gfx950 correction, MI355X_MICROARCH.md) in hbm_bytes_per_launch_corrected"}
for k in sorted(set(fe) | set(wr)):
    f, w = fe.get(k, []), wr.get(k, [])
    f, w = f[len(f) // 2:], w[len(w) // 2:]
    if not f or not w: continue
    out[k] = {"FETCH_SIZE_KiB_mean": sum(f) / len(f), "WRITE_SIZE_KiB_mean": sum(w) / len(w), "dispatches": len(fe[k]),
              "hbm_bytes_per_launch_corrected": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024}
json.dump(out, open(os.path.join(dst, "jit_pmc_summary.json"), "w"), indent=1)
print(open(os.path.join(dst, "jit_pmc_summary.json")).read()[:1500])
