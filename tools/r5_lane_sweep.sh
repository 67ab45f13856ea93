#!/bin/bash
# GPU box: the headline run (configs[2], mask on) over frames per step and lanes -- "batch lanes" pairs; value and ms per step of a short run each.
# Round 5 final kernels: 128/2 1570, 192/3 1580, 96/3 1555, 256/2 1376 (128-frame launches pass the 2 GiB input limit of the Winograd kernel), 192/2 1584, 128/1 1341.
for cfg in "128 2" "192 3" "96 3" "256 2" "192 2" "128 1"; do set -- $cfg; timeout -k 10 200 python bench.py --batch $1 --streams $2 --steps 12 --warmup 3 --leg-steps 0 --cpu-frames 0 --cpu-cores 0 --latency-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 lanes $2:', d['value'], d['ms_per_step'])"; done
