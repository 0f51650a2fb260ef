#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output dir (gpurun_out/prof_<tag>) into profiles/:
   <tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats summary, verbatim)
   <tag>_summary.json      (per-kernel mean duration + HBM traffic per launch from the PMC passes)
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B?
rocprofv3 reports them in KB (x1024 B); on gfx950 FETCH_SIZE counts half the bytes of wide
coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
out = {"tag": tag, "kernels": {}}
for r in csv.DictReader(open(stats)):
    out["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                 "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
def pmc(kind, counter):
    f = glob.glob(os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv"))
    if not f:
        return None
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
traffic = {}
for name in out["kernels"]:
    if fetch and name in fetch and write and name in write:
        fb = [2.0 * 1024.0 * v for v in fetch[name]]
        wb = [1024.0 * v for v in write[name]]
        traffic[name] = {"launches": len(fb), "fetch_bytes_per_launch_corrected": sum(fb) / len(fb),
                         "write_bytes_per_launch": sum(wb) / len(wb),
                         "hbm_bytes_per_launch": sum(fb) / len(fb) + sum(wb) / len(wb),
                         "per_launch_fetch_raw_kb": fetch[name][:8], "per_launch_write_raw_kb": write[name][:8]}
out["traffic"] = traffic
out["bench_line"] = open(os.path.join(src, "bench_trace.json")).read().strip()
json.dump(out, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line"}, indent=1)[:3000])
