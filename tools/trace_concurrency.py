#!/usr/bin/env python3
"""How many kernels run at once in a rocprofv3 --kernel-trace of the bench: idle share, mean concurrency, and the
share of wall time each kernel is resident (tools/trace_concurrency.py DIR)."""
import collections
import csv
import glob
import os
import sys
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("amos::", "")) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[len(rows) // 4][0]          # skip warm-up: analyse the last three quarters
rows = [r for r in rows if r[0] >= t0]
ev = []
for s, e, n in rows:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
span = ev[-1][0] - ev[0][0]
busy = conc = 0
active = 0
res = collections.Counter()
cur = collections.Counter()
last = ev[0][0]
hist = collections.Counter()
for t, d, n in ev:
    dt = t - last
    if active > 0:
        busy += dt
    conc += active * dt
    hist[min(active, 6)] += dt
    for k, v in cur.items():
        if v > 0:
            res[k] += dt
    active += d
    cur[n] += d
    last = t
print(f"span {span/1e6:.2f} ms  busy {100*busy/span:.1f} %  mean concurrency {conc/span:.2f}")
print("time share by number of kernels resident:", {k: round(100 * v / span, 1) for k, v in sorted(hist.items())})
for k, v in res.most_common(12):
    print(f"  {k:28s} resident {100*v/span:5.1f} % of wall time")
