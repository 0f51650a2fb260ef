#!/usr/bin/env python3
"""One frame of the bench's frame loop out of a rocprofv3 kernel trace: every launch in start order with its gap to the end of
what ran before, its duration, and the frame's span / busy time (overlapping launches counted once).
usage: python tools/frame_timeline.py <..._kernel_trace.csv> [frame index from the end, default 5]
(frames are cut at the live volume's multi-view sweep, integrate_depth_multi_column_kernel; the last three frames of a run are
the stage-timed ones, with a device synchronisation after every stage: take an earlier one)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "integrate_depth_multi_column_kernel" in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
end = t0
busy = 0
print("# frame %d from the end: %d launches, span %.1f us" % (back, i1 - i0, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
print("#   start us    gap us   dur us  kernel")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("dfh::", "")
    name = name.split("(")[0][:70]
    gap = (s - end) / 1e3
    print("%11.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print("# busy %.1f us (overlapping launches counted once)" % (busy / 1e3))
