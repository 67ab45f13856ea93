#!/bin/bash
# Runs on the GPU box: several rocprofv3 --pmc passes of a single-lane bench for kernel analysis.
# Output: gpurun_out/deep/<pass>/..., summary printed by tools/pmc_summary.py (raw averages per launch).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/deep
rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --cpu-frames 0 --streams 1 ${BENCH_EXTRA:-}"
i=0
for set in "${@}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $ARGS > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; exit 1; }
  echo "== pass $i: $set"
  python3 $R/tools/pmc_summary.py $O/p$i raw
done
