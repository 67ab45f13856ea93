#!/bin/bash
# GPU box: alternate the in-tree library (A) and build/var_$1/libamos_frontend.so (B), N rounds, 2-lane bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  unset AMOS_FRONTEND_LIB
  timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 3 --cpu-frames 0 "$@" 2>/dev/null | python3 $R/tools/show_bench.py A | head -1
  AMOS_FRONTEND_LIB=$R/build/var_$V/libamos_frontend.so timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 3 --cpu-frames 0 "$@" 2>/dev/null | python3 $R/tools/show_bench.py B_$V | head -1
done
