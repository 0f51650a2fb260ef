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
stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getsize)      # (the bench process, not a child)
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
def traffic_of(fetch, write, names):
    traffic = {}
    for name in names:
        if fetch and name in fetch and write and name in write:
            fb = [2.0 * 1024.0 * v for v in fetch[name]]
            wb = [1024.0 * v for v in write[name]]
            traffic[name] = {"launches": len(fb), "fetch_bytes_per_launch_corrected": sum(fb) / len(fb),
                             "write_bytes_per_launch": sum(wb) / len(wb),
                             "hbm_bytes_per_launch": sum(fb) / len(fb) + sum(wb) / len(wb),
                             "max_hbm_bytes_of_a_launch": max(a + b for a, b in zip(fb, wb)) if len(fb) == len(wb) else None,
                             "per_launch_fetch_raw_kb": fetch[name][:8], "per_launch_write_raw_kb": write[name][:8]}
    return traffic
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
out["traffic"] = traffic_of(fetch, write, sorted(set(fetch or {}) & set(write or {})))
# side runs of the same round (tools/profile_round.sh): K2 / K3 (kbench_fv.py) and the 512^3 x 8 multi-view sweep
def pmc_dir(d, counter):
    f = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not f:
        return None
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
for side in ("fv", "mv"):
    f_, w_ = pmc_dir(side + "_FETCH_SIZE", "FETCH_SIZE"), pmc_dir(side + "_WRITE_SIZE", "WRITE_SIZE")
    if f_ and w_:
        out["traffic_" + side] = traffic_of(f_, w_, sorted(set(f_) & set(w_)))
    st = glob.glob(os.path.join(src, side + "_trace", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(dst, tag + "_" + side + "_kernel_stats.csv"))
        out["kernels_" + side] = {r["Name"]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
                                  for r in csv.DictReader(open(st[0]))}
    txt = os.path.join(src, side + "_trace.txt")
    if os.path.exists(txt):
        out[side + "_output"] = [l.strip() for l in open(txt) if "us" in l and "amdgpu.ids" not in l]
out["bench_line"] = open(os.path.join(src, "bench_trace.json")).read().strip()
json.dump(out, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line"}, indent=1)[:3000])
