import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fstart = [i for i, r in enumerate(rows) if "depth_pyramid" in r["Kernel_Name"]][-3:]
seg = rows[fstart[0]:fstart[1]]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
gn = 0
for r in seg:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    k = k[:90]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "gn_build_data" in k: gn += 1
    if gn > 1 and gn < 10 and any(x in k for x in ("gn_build", "gn_gather", "pcg_cg1", "fillBuffer")):
        prev_end = e
        continue
    print("%8.1f  gap %6.1f  dur %7.1f  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, k[-70:], r.get("Grid_Size", "")))
    prev_end = e
