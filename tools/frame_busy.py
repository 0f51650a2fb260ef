#!/usr/bin/env python3
"""GPU-busy time against wall time of steady-state frames: run under `rocprofv3 --kernel-trace` (tools/kbench_frames.py prints a
marker kernel-free gap between frames; here the trace of the last frames is summed).  usage: python tools/frame_busy.py <trace.csv> <n_frames>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames are delimited by the multi-view live-volume sweep (first kernel of SlabFrame.step after the two fills)
starts = [i for i, r in enumerate(rows) if "integrate_depth_multi" in r["Kernel_Name"] or "depth_pyramid" in r["Kernel_Name"]]
# keep the first kernel of every frame: a pyramid launch followed by classify + multi sweep
fstart = [i for i in starts if "depth_pyramid" in rows[i]["Kernel_Name"]][-nf:]
out = []
for a, b in zip(fstart[:-1], fstart[1:]):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    by = {}
    for r in seg:
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")[-40:]
        by[k] = by.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out.append(((t1 - t0) / 1e6, busy / 1e6, len(seg), by))
for w, b, n, by in out[-4:]:
    print("frame: wall %.3f ms, GPU busy %.3f ms (%.0f %%), %d launches" % (w, b, 100 * b / w, n))
w, b, n, by = out[-1]
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-42s %.3f ms" % (k, v / 1e6))
# idle gaps of the last frame: where the device waits for the host
seg = rows[fstart[-2]:fstart[-1] + 1]
gaps = []
for a, b in zip(seg[:-1], seg[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 8.0:
        short = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")[-34:]
        gaps.append((g, short(a), short(b)))
print("idle gaps > 8 us in the last frame: %d, total %.0f us" % (len(gaps), sum(g for g, _, _ in gaps)))
for g, a, b in gaps:
    print("   %6.1f us after %-36s before %s" % (g, a, b))
small = sum(max(0.0, (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3) for a, b in zip(seg[:-1], seg[1:])) - sum(g for g, _, _ in gaps)
print("   all other gaps together: %.0f us over %d launches" % (small, len(seg) - 1))
