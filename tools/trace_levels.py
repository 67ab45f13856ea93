#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace csv, grouped by position in
the repeating launch sequence (e.g. the 7 k_pyramid_level launches of a pass): tools/trace_levels.py DIR NAME PERIOD"""
import csv
import glob
import sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
name, period = sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
n = len(d) // period
for k in range(period):
    v = d[k::period][1:]
    gp = gaps[k::period][1:]
    print(f"{name}[{k}] n={len(v)} avg {sum(v)/max(len(v),1):8.2f} us  min {min(v):8.2f}  gap-after avg {sum(gp)/max(len(gp),1):8.2f} us  grid {rows[k]['Grid_Size_X'] if 'Grid_Size_X' in rows[k] else ''}")
