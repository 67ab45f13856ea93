#!/bin/bash
# GPU box: time and L2-miss traffic (FETCH_SIZE) of the F(2 x 4) kernel for the group sizes built by tools/w24_variants.sh (G<n>; base = in-tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/w24_map
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/amos-slam_amd/csrc/build/libamos_frontend_w24_$v.so; fi
  echo -n "$v: "; timeout -k 10 120 python3 $R/tools/winograd_probe.py --f24-time 2>/dev/null | tail -1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$v -- python3 $R/tools/winograd_one.py > $O/$v.log 2>&1 || { echo "pmc failed"; continue; }
  python3 $R/tools/pmc_summary.py $O/$v raw | grep -i "24_conv"
done
