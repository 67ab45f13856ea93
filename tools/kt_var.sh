#!/bin/bash
# GPU box: kernel-trace the pyramid kernels for each experimental library under build/var_*/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  rm -rf $R/gpurun_out/kt_$v
  if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/build/var_$v/libamos_frontend.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$v -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-frames 0 --streams 1 > $R/gpurun_out/kt_$v.log 2>&1 || { tail -5 $R/gpurun_out/kt_$v.log; exit 1; }
  echo "== $v"
  python3 $R/tools/trace_levels.py $R/gpurun_out/kt_$v "${KT_KERNEL:-k_pyramid_level<}" ${KT_PERIOD:-7}
done
