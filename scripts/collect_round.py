#!/usr/bin/env python
"""gpurun_out/<tag> (scratch) -> profiles/<tag> (committed): the small files of scripts/gpu_round2.sh plus the rocprofv3
summaries of the profiled bench command (dominant kernel: ggrs_jit_tick).

usage: collect_round.py <tag>"""
import collections, csv, glob, json, os, shutil, subprocess, sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "*")):
    if os.path.isfile(f) and os.path.getsize(f) < 2 << 20 and not f.endswith((".err", ".log")) or f.endswith(("pytest_gpu.log", "smoke.log")):
        shutil.copy(f, dst)
for name, sub in (("kernel_stats.csv", "prof_stats"), ("kernel_stats_allhot.csv", "prof_stats_allhot")):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits: shutil.copy(hits[0], os.path.join(dst, name))
subprocess.check_call([sys.executable, "scripts/pmc_summary.py", src, dst], stdout=subprocess.DEVNULL)
if glob.glob(os.path.join(src, "prof_allhot_fetch", "**", "*counter_collection.csv"), recursive=True):
    subprocess.check_call([sys.executable, "scripts/pmc_summary.py", src, dst, "allhot", "prof_allhot"], stdout=subprocess.DEVNULL)

# the SQ / GRBM pass of the same command: VALU occupancy of the dominant kernel and the clock it ran at
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "prof_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = {}
for k, cs in agg.items():
    if "tick" not in k: continue
    sq[k] = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in cs.items()}
    sq[k]["dispatches"] = max(len(v) for v in cs.values())
if sq:
    json.dump({"note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY of the bench command; means over the "
                       "second half of the dispatches; SQ_* cycle counters are quad-cycles summed over all SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs",
               "kernels": sq}, open(os.path.join(dst, "sq_summary.json"), "w"), indent=1)

# the ALU ceiling bench.py's --config 5 line is priced against
ub = os.path.join(src, "ubench_alu.txt")
if os.path.exists(ub):
    best = 0.0
    for line in open(ub):
        if line.startswith("diffuse"):
            best = max(best, float(line.split()[-2]))
    if best:
        json.dump({"diffuse_G_per_s": best, "source": f"profiles/{tag}/ubench_alu.txt (scripts/ubench_alu.hip: SeaHash diffuse chains on every CU; best of 1-8 chains x 1-8 waves per SIMD)"},
                  open(os.path.join("profiles", "alu_ceiling.json"), "w"), indent=1)
print(open(os.path.join(dst, "pmc_summary.json")).read()[:1500])
