#!/bin/bash
# Runs on the GPU box (through gpurun): bench JSON, rocprofv3 kernel stats and the two PMC passes of the
# same command; outputs under gpurun_out/final/.  tools/make_profile_summary.py turns them into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O
mkdir -p $O
timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 3 --check > $O/bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 > $O/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 > $O/write.log 2>&1 || exit 1
tail -1 $O/bench.log | python3 $R/tools/show_bench.py final
