#!/bin/bash
# GPU box: headline (mask on) frames/s for lane counts / frames per lane.  Output: gpurun_out/sweep/*.json and a table.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sweep
rm -rf $O; mkdir -p $O
IFS=";" read -ra CFGS <<< "${SWEEP:-1 32;2 64;3 96;4 128;2 96;3 144;2 128}"
for cfg in "${CFGS[@]}"; do
  IFS=" " read -r a b <<< "$cfg"; set -- $a $b
  timeout -k 10 200 python3 $R/bench.py --streams $1 --batch $2 --steps 5 --warmup 2 --leg-steps 0 --cpu-frames 0 > $O/s$1_b$2.json 2> $O/s$1_b$2.err || { echo "streams $1 batch $2 failed"; tail -3 $O/s$1_b$2.err; continue; }
  python3 -c "
import json
d=json.loads(open('$O/s$1_b$2.json').read().strip().splitlines()[-1]); print('streams $1 batch $2:', d['value'], 'frames/s', d['roofline_mask']['achieved'], 'TF', d['stage_ms_per_launch'].get('mask_pass'))"
done
