#!/bin/bash
# GPU box: kernel trace of ONE mask lane alone (bench.py --streams 1 --batch 32) and its summary by kernel kind.
#   tools/one_lane_trace.sh <tag>   -> gpurun_out/lane_<tag>/ and gpurun_out/lane_<tag>.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lane_$1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --streams 1 --batch 32 --steps 3 --warmup 2 --leg-steps 0 --cpu-frames 0 > $O/run.log 2>&1 || { echo "trace failed"; tail -5 $O/run.log; exit 1; }
python3 $R/tools/one_lane_mask_trace.py $O > $R/gpurun_out/lane_$1.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
cat $R/gpurun_out/lane_$1.txt
