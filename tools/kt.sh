#!/bin/bash
# GPU box: kernel trace of a single-lane bench and per-launch durations of the main kernels.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-frames 0 --streams 1 ${BENCH_EXTRA:-} > $R/gpurun_out/kt.log 2>&1 || { tail -5 $R/gpurun_out/kt.log; exit 1; }
cd $R
python3 tools/trace_levels.py gpurun_out/kt k_pyramid_level0 1
python3 tools/trace_levels.py gpurun_out/kt 'k_pyramid_level<' 7
for k in k_fast k_blur k_octree k_orient k_describe k_bf_best2; do python3 tools/trace_levels.py gpurun_out/kt $k 1; done
