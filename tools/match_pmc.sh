#!/bin/bash
# GPU box: counters of the two brute-force kernels on the resident 128-frame workload (tools/match_bench.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/mpmc
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $O/counters.txt | sort -u | head -20
i=0
for set in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/match_bench.py > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/p$i/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_bf_best2" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "(1e6 per launch)")
PY
done
