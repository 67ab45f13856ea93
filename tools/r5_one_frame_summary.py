"""From a rocprofv3 kernel trace of tools/r5_one_frame_run.py: the LAST pass (from one k_mask_pre_a to the next): wall time, time with a kernel
running, launches, kernels by kind, then every kernel in order (start offset, duration, gap before it, grid, name)."""
import collections
import csv
import glob
import os
import sys

trace = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_mask_pre_a" in r["Kernel_Name"]]
rs = rows[idx[-2]:idx[-1]]
span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
busy, end = 0, 0
for r in rs:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > end:
        busy += e - max(s, end)
        end = e


def kind(n):
    if "amos::" in n:
        return "amos::" + n.split("amos::")[1].split("(")[0][:48]
    if "igemm" in n or "ck" in n[:40] or "Cijk" in n or "gemm" in n.lower() or "conv" in n.lower() or "SubTensor" in n or "naive" in n.lower() or "batched_transpose" in n:
        return "library: " + n[:60]
    return n.replace("void at::native::", "")[:70]


acc = collections.defaultdict(lambda: [0, 0])
for r in rs:
    acc[kind(r["Kernel_Name"])][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[kind(r["Kernel_Name"])][1] += 1
print("one pass: %.3f ms wall, %.3f ms with a kernel running (%.1f %%), %d launches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rs)))
for g, (d, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%8.1f us %5.1f %% %4d  %s" % (d / 1e3, 100.0 * d / span, c, g))
print()
t0, prev = int(rs[0]["Start_Timestamp"]), int(rs[0]["Start_Timestamp"])
for r in rs:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = r.get("Workgroup_Size", "?")
    print("%8.1f us %7.1f us  gap %6.1f  grid %-9s wg %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size", "?"), wg,
                                                                   r["Kernel_Name"].replace("void at::native::", "")[:110]))
    prev = max(prev, e)
