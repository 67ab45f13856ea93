#!/bin/bash
# GPU box: kernel trace of ONE mask lane alone at 64 frames per forward (the bench's launch size), summary by kernel kind and every launch >= 40 us in order
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lane_r4_64
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --streams 1 --batch 64 --steps 3 --warmup 2 --leg-steps 0 --cpu-frames 0 > $O/run.log 2>&1 || { echo "trace failed"; tail -5 $O/run.log; exit 1; }
python3 $R/tools/one_lane_mask_trace.py $O > $R/gpurun_out/lane_r4_64.txt
python3 $R/tools/one_lane_sequence.py $O 40 > $R/gpurun_out/lane_r4_64_sequence.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
head -14 $R/gpurun_out/lane_r4_64.txt
