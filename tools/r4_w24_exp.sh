#!/bin/bash
# GPU box: time of the F(2 x 4) kernel variants built by tools/w24_variants.sh on four layer shapes (tools/winograd_probe.py --f24-time)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB="$R/amos-slam_amd/csrc/build/libamos_frontend_w24_$v.so"; fi
  echo -n "$v: "; timeout -k 10 120 python3 $R/tools/winograd_probe.py --f24-time 2>/dev/null | tail -1
done
