#!/bin/bash
# GPU box: lanes x frames-per-forward sweep of the default (mask on) bench: "<lanes> <batch>" pairs; frames per forward = batch / lanes.
R=${GRAFT_REPO_ROOT:-/root/repo}
while [ $# -ge 2 ]; do
  s=$1; b=$2; shift 2
  timeout -k 10 300 python3 $R/bench.py --streams $s --batch $b --steps 10 --warmup 2 --leg-steps 0 --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); m=d['roofline_mask']
        print('lanes $s batch $b:', d['value'], 'fps', d['ms_per_step'], 'ms/step, lane pass', m['lane_pass_ms'], 'ms, executed TF', m['achieved'], 'frac', m['frac'])
"
done
