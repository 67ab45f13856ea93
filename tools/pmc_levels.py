#!/usr/bin/env python3
"""rocprofv3 --pmc csv -> counter values per position in a kernel's repeating launch sequence:
tools/pmc_levels.py DIR KERNEL_SUBSTRING PERIOD"""
import collections
import csv
import glob
import sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
name, period = sys.argv[2], int(sys.argv[3])
by_disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if name in r["Kernel_Name"]:
        by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
disp = [by_disp[k] for k in sorted(by_disp)]
for k in range(period):
    rows = disp[k::period][1:] or disp[k::period]
    keys = sorted(rows[0])
    print(f"{name}[{k}]", {c: round(sum(r[c] for r in rows) / len(rows), 1) for c in keys})
