#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv per kernel (helper for profiling runs)."""
import collections
import csv
import glob
import sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "amos::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
raw = len(sys.argv) > 2 and sys.argv[2] == "raw"
for k, v in acc.items():
    if raw:
        print(k, {c: round(sum(x) / len(x), 2) for c, x in v.items()}, "(avg/launch)")
    else:
        print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(1e6, avg/launch)")
