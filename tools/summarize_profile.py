#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_gpu.sh) into a summary: per-kernel time stats and
per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE passes, corrected with the calibration
factors measured on the known-byte stream-copy kernels of the same run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ising10m")


def find(d, pat):
    return sorted(glob.glob(os.path.join(out, d, "**", pat), recursive=True))


def kernel_stats():
    rows = []
    for fn in find("trace", "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(fn)))
    return rows


def counters(d):
    """{kernel name: {counter: [values per dispatch]}}"""
    res = defaultdict(lambda: defaultdict(list))
    for fn in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res


def mean(x):
    return sum(x) / len(x) if x else float("nan")


summary = {"workload": wl, "kernels": [], "pmc": {}, "calibration": {}}
print("== kernel time stats (rocprofv3 --kernel-trace --stats) ==")
for r in kernel_stats():
    name = r.get("Name", "")
    short = name.split("(")[0][:60]
    print("%-62s calls=%6s avg_ns=%12s pct=%6s" % (short, r.get("Calls"), r.get("AverageNs"), r.get("Percentage")))
    summary["kernels"].append({"name": short, "calls": int(r.get("Calls", 0)),
                               "avg_ns": float(r.get("AverageNs", 0)),
                               "total_ns": float(r.get("TotalDurationNs", 0)),
                               "pct": float(r.get("Percentage", 0))})

NB = float(1 << 30)
cal = {}
for tag, d in (("FETCH_SIZE", "cal_fetch"), ("WRITE_SIZE", "cal_write")):
    for k, c in counters(d).items():
        if "k_stream_copy" in k and tag in c:
            width = 16 if ("uint4" in k or "HIP_vector_type" in k) else 4
            v = mean(c[tag])
            # counter unit: the guide quotes bytes = value * 1024 (KB); factor = true / reported
            cal[(tag, width)] = NB / (v * 1024.0) if v else float("nan")
            print("calibration %s width=%d: counter=%.1f -> true/reported(KB) = %.3f" % (tag, width, v, cal[(tag, width)]))
summary["calibration"] = {"%s_w%d" % k: v for k, v in cal.items()}

print("== PMC per launch ==")
traffic = {}
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for k, c in counters(d).items():
        if "k_gibbs" in k or "k_learn" in k:
            short = k.split("(")[0][:40]
            for cn, vals in c.items():
                print("%-42s %-22s mean=%.4g n=%d" % (short, cn, mean(vals), len(vals)))
                summary["pmc"].setdefault(short, {})[cn] = mean(vals)
for k, c in summary["pmc"].items():
    f = c.get("FETCH_SIZE")
    w = c.get("WRITE_SIZE")
    if f is not None and w is not None:
        # dword-wide accesses: use the width-4 calibration of the same run
        fb = f * 1024.0 * cal.get(("FETCH_SIZE", 4), 1.0)
        wb = w * 1024.0 * cal.get(("WRITE_SIZE", 4), 1.0)
        c["hbm_read_bytes_per_launch"] = fb
        c["hbm_write_bytes_per_launch"] = wb
        c["hbm_bytes_per_launch"] = fb + wb
        print("%s: HBM bytes/launch = %.4g (read %.4g + write %.4g), corrected" % (k, fb + wb, fb, wb))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
# bench.py reports roofline.traffic from profiles/traffic.json: HBM bytes per launch of the dominant kernel
# dominant sweep kernel = the gibbs/learn kernel with the largest total time in the stats pass
dom, best = None, -1.0
for kr in summary["kernels"]:
    if not ("k_gibbs" in kr["name"] or "k_learn" in kr["name"]):
        continue
    for k, c in summary["pmc"].items():
        if k[:36] == kr["name"][:36] and "hbm_bytes_per_launch" in c and kr["total_ns"] > best:
            dom, best = c["hbm_bytes_per_launch"], kr["total_ns"]
            summary["dominant_kernel"] = kr["name"]
# issue-side figures of the dominant kernel (bench.py prints them as roofline.issue)
domk = None
for kr in sorted(summary["kernels"], key=lambda k: -k["total_ns"]):
    if "k_gibbs" in kr["name"] or "k_learn" in kr["name"]:
        domk = kr
        break
if domk is not None:
    for k, c in summary["pmc"].items():
        if k[:36] == domk["name"][:36] and "SQ_INSTS_VALU" in c:
            wc = c.get("SQ_WAVE_CYCLES") or float("nan")
            issue = {"kernel": domk["name"], "avg_launch_us_profiled": domk["avg_ns"] / 1e3,
                     "valu_insts_per_launch": c["SQ_INSTS_VALU"], "waves_per_launch": c.get("SQ_WAVES"),
                     "active_inst_any_frac_of_wave_cycles": c.get("SQ_ACTIVE_INST_ANY", float("nan")) / wc,
                     "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", float("nan")) / wc,
                     "vmem_rd_insts_per_launch": c.get("SQ_INSTS_VMEM_RD")}
            json.dump({wl: issue}, open(os.path.join(out, "issue_%s.json" % wl), "w"))
            print("issue side:", issue)
            break
if dom is not None:
    json.dump({wl: dom}, open(os.path.join(out, "traffic_%s.json" % wl), "w"))
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print("dominant kernel:", summary.get("dominant_kernel"), "HBM bytes/launch", dom)
