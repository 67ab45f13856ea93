#!/bin/bash
# GPU box: bench.py --config c5 (1920x1080, 12 levels, 4000 features) over library variants, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/amos-slam_amd/csrc/build/libamos_frontend_$v.so; fi
    timeout -k 10 300 python3 $R/bench.py --config c5 --steps 10 --warmup 2 --cpu-frames 0 2>/dev/null | python3 $R/tools/show_bench.py $v | head -1 | cut -c1-260
  done
done
