#!/bin/bash
# GPU box: hardware-queue / lane-count sweep of `bench.py --config c2` (in-tree library): "<queues> <lanes> <batch>" triples.
R=${GRAFT_REPO_ROOT:-/root/repo}
while [ $# -ge 3 ]; do
  q=$1; s=$2; b=$3; shift 3
  for rep in 1 2; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 $R/bench.py --config c2 --steps 40 --warmup 3 --cpu-frames 0 --streams $s --batch $b 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('queues $q lanes $s batch $b:', d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k, v) for k, v in d['stage_ms_per_launch'].items()))
"
  done
done
