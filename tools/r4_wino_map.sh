#!/bin/bash
# GPU box: time (tools/winograd_probe.py --big-only) and L2-miss traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/winograd_one.py) of the
# Winograd kernel's work-group -> (m block, n tile) map variants built by tools/wino_variants.sh (M<map>G<group>).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wino_map
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export AMOS_FRONTEND_LIB=$R/amos-slam_amd/csrc/build/libamos_frontend_$v.so
  echo "== $v" | tee -a $O/summary.txt
  timeout -k 10 120 python3 $R/tools/winograd_probe.py --big-only 2>&1 | tail -3 | tee -a $O/summary.txt || exit 1
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$v-$set -- python3 $R/tools/winograd_one.py > $O/$v-$set.log 2>&1 || { echo "pmc pass failed"; tail -5 $O/$v-$set.log; exit 1; }
    python3 $R/tools/pmc_summary.py $O/$v-$set raw | grep -i wino_conv | tee -a $O/summary.txt
  done
done
