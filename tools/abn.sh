#!/bin/bash
# GPU box: bench the in-tree library and each build/var_<name> library, N rounds interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/build/var_$v/libamos_frontend.so; fi
    timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 3 --cpu-frames 0 ${BENCH_EXTRA:-} 2>/dev/null | python3 $R/tools/show_bench.py $v | head -1
  done
done
