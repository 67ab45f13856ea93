#!/bin/bash
# GPU box: PMC passes over the Winograd kernel alone (tools/winograd_one.py: the 138 x 138 protonet layer, 32 frames, 4 launches)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wino_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/winograd_one.py > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; exit 1; }
  echo "== pass $i: $set"
  python3 $R/tools/pmc_summary.py $O/p$i raw | grep -i wino
done
