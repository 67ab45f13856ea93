#!/bin/bash
# GPU box: kernel trace of the one-frame mask pass.   tools/r5_one_frame_trace.sh <tag> [eager]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/one_frame_$1
rm -rf $O; mkdir -p $O
[ "$2" = eager ] && export AMOS_ONE_FRAME_EAGER=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/r5_one_frame_run.py 20 > $O/run.log 2>&1 || { echo "trace failed"; tail -5 $O/run.log; exit 1; }
tail -2 $O/run.log
python3 $R/tools/r5_one_frame_summary.py $O > $R/gpurun_out/one_frame_$1.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
head -60 $R/gpurun_out/one_frame_$1.txt
