"""One lane alone (bench.py --streams 1 --batch 32): from a rocprofv3 kernel trace, the last timed step's wall time, its kernel
time, the idle time between kernels and the kernels grouped by kind.   python tools/one_lane_mask_trace.py <trace dir>"""
import collections
import csv
import glob
import os
import sys

trace = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_import_color_mask" in r["Kernel_Name"]]
rs = rows[idx[-2]:idx[-1]]
span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
busy, end = 0, 0
for r in rs:  # union of the kernel intervals
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > end:
        busy += e - max(s, end)
        end = e
acc = collections.defaultdict(lambda: [0, 0])
for r in rs:
    n = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "k_conv1x1" in n:
        g = "amos conv1x1 GEMM"
    elif "igemm" in n or "ck16" in n or "conv" in n.lower() and "amos" not in n or "Cijk" in n:
        g = "MIOpen / rocBLAS conv+gemm"
    elif "amos::" in n:
        g = "amos::" + n.split("amos::")[1].split("(")[0][:40]
    else:
        g = n.replace("void at::native::", "")[:90]
    acc[g][0] += d
    acc[g][1] += 1
print("step: %.3f ms wall, %.3f ms with a kernel running (%.1f %%), %d launches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rs)))
for g, (d, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%7.3f ms %5.1f %% %5d  %s" % (d / 1e6, 100.0 * d / span, c, g))
