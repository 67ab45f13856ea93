#!/bin/bash
# GPU box: single-frame latency of the drop-in API and the per-kernel durations at batch 1 (kernel trace).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lat
rm -rf $O; mkdir -p $O
python3 $R/tools/latency.py
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/tools/latency.py > $O/t.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one frame = from one k_pyramid_level0_wide to the next
idx = [i for i, r in enumerate(rows) if "k_pyramid_level0_wide" in r["Kernel_Name"]]
a, b = idx[30], idx[31]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} us +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:6.1f}  {r["Kernel_Name"][:60]}')
print("frame period us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
PY
