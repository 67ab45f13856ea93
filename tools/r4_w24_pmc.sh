#!/bin/bash
# GPU box: counter passes over the F(2 x 4) Winograd kernel alone (tools/winograd_one.py: proto_net 256 -> 256 at 138 x 138, 32 frames, 4 launches)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/w24_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/winograd_one.py > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log | cut -c1-300; continue; }
  echo "== pass $i: $set"
  python3 $R/tools/pmc_summary.py $O/p$i raw | grep -i wino | cut -c1-400
done
