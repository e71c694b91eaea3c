#!/usr/bin/env python
"""Run a command under several `rocprofv3 --pmc` passes (counters validated against `rocprofv3 -L`, <= 4 TCC-block and
<= 8 SQ-block counters per pass, never combined with tracing flags) and summarise per kernel.

usage: pmc_passes.py <outdir> <counters.txt from rocprofv3 -L> -- <command ...>
Writes <outdir>/pmc_<k>/ (raw csv) and <outdir>/pmc_counters.json:
  {kernel: {counter: mean over the second half of its dispatches, "dispatches": n, "dur_us_mean": ...}}
"""
import collections, csv, glob, json, os, re, subprocess, sys

WISH_TCC = [
    "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WR_UNCACHED_32B_sum",
    "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_sum",
    "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum",
    "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_WRITE_sum",
    "TCC_WRITEBACK_sum", "TCC_NORMAL_WRITEBACK_sum", "TCC_NORMAL_EVICT_sum", "TCC_TAG_STALL_sum",
    "TCC_STREAMING_REQ_sum", "TCC_NC_REQ_sum", "TCC_UC_REQ_sum", "TCC_CC_REQ_sum",
    "TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_RDREQ_LEVEL_sum",
    "TCC_EA0_ATOMIC_sum", "TCC_ATOMIC_sum", "TCC_SRC_FIFO_FULL_sum", "TCC_LATENCY_FIFO_FULL_sum",
]
WISH_SQ = [
    "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD",
    "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
]
WISH_OTHER = ["GRBM_GUI_ACTIVE", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TA_TCP_STATE_READ_sum",
              "TA_BUSY_avr", "TA_TA_BUSY_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_LATENCY_sum"]


def main():
    out, listing = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    txt = open(listing).read()
    have = lambda n: re.search(r"\b" + re.escape(n) + r"\b", txt) is not None
    wish_sq = WISH_SQ + [c for c in os.environ.get("PMC_SQ_EXTRA", "").split(",") if c]      # more SQ-block counters for one study
    only_sq = os.environ.get("PMC_ONLY_SQ") == "1"
    tcc = [] if only_sq else [c for c in WISH_TCC if have(c)]
    sq = [c for c in wish_sq if have(c)]
    oth = [c for c in (["GRBM_GUI_ACTIVE"] if only_sq else WISH_OTHER) if have(c)]
    missing = [c for c in WISH_TCC + WISH_SQ + WISH_OTHER if not have(c)]
    passes = []
    while tcc or sq or oth:
        p = tcc[:4] + sq[:6] + oth[:2]
        tcc, sq, oth = tcc[4:], sq[6:], oth[2:]
        passes.append(p)
    max_passes = int(os.environ.get("PMC_MAX_PASSES", "10"))
    passes = passes[:max_passes]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    log = {"passes": passes, "missing_counters": missing, "failed": []}
    for k, p in enumerate(passes):
        d = os.path.join(out, f"pmc_{k}")
        r = subprocess.run(["rocprofv3", "--pmc", *p, "-f", "csv", "-d", d, "-o", "pmc", "--"] + cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        open(os.path.join(out, f"pmc_{k}.log"), "w").write(r.stdout[-4000:])
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode or not files:
            log["failed"].append({"pass": k, "rc": r.returncode, "counters": p})
            continue
        for f in files:
            for row in csv.DictReader(open(f)):
                kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[kn][row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
    summ = {}
    for kn, cs in agg.items():
        d = {}
        for c, vals in cs.items():
            vals.sort()
            v = [x for _, x in vals]
            d["dispatches"] = len(v)
            v = v[len(v) // 2:]
            d[c] = sum(v) / len(v)
        summ[kn] = d
    log["kernels"] = summ
    json.dump(log, open(os.path.join(out, "pmc_counters.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in log.items() if k != "kernels"}, indent=1))
    for kn, d in summ.items():
        if "tick" in kn or "copy" in kn:
            print(kn, json.dumps(d, indent=1) if not only_sq else json.dumps({k: round(v) for k, v in d.items()}))


if __name__ == "__main__":
    main()
