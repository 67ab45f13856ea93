#!/bin/bash
# GPU box: N interleaved rounds of `bench.py --config c2` over library variants (amos-slam_amd/csrc/build/libamos_frontend_<name>.so;
# "base" = the in-tree library).  One line per run: value, ms per step, stage times (4 lanes), then lane-alone stage times.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/amos-slam_amd/csrc/build/libamos_frontend_$v.so; fi
    timeout -k 10 200 python3 $R/bench.py --config c2 --steps 40 --warmup 3 --cpu-frames 0 ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d.get('stage_ms_per_launch_lane_alone') or {}
        print('$v', d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k, v) for k, v in d['stage_ms_per_launch'].items()), '| alone', ' '.join('%s=%.3f' % (k, v) for k, v in a.items()))
"
  done
done
