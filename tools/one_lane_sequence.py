"""Ordered kernel list of the last step of a one-lane trace (see tools/one_lane_mask_trace.py): start offset, duration, name, for kernels
of at least <min us>.   python tools/one_lane_sequence.py <trace dir> [min_us]"""
import csv
import glob
import os
import sys

trace = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_import_color_mask" in r["Kernel_Name"]]
rs = rows[idx[-2]:idx[-1]]
t0 = int(rs[0]["Start_Timestamp"])
for r in rs:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= min_us:
        n = r["Kernel_Name"].replace("void at::native::", "").replace("(anonymous namespace)::", "")
        print("%9.3f ms %8.1f us  grid %-10s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Grid_Size", "?"), n[:150]))
