#!/bin/bash
# GPU box: A/B of the current library against build/libamos_frontend_r01.so (round-1 kernels):
# per-stage times (one lane alone and the 4-lane default) and VALU instruction counts per kernel.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab
rm -rf $O; mkdir -p $O
for tag in new r01; do
  if [ $tag = r01 ]; then export AMOS_FRONTEND_LIB=$R/build/libamos_frontend_r01.so; else unset AMOS_FRONTEND_LIB; fi
  timeout -k 10 120 python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 30 --warmup 3 --cpu-frames 0 > $O/${tag}_1lane.json 2> $O/${tag}_1lane.err || { echo "$tag 1lane failed"; tail -5 $O/${tag}_1lane.err; exit 1; }
  timeout -k 10 120 python3 $R/bench.py --config c2 --steps 100 --warmup 5 --cpu-frames 0 > $O/${tag}_4lane.json 2> $O/${tag}_4lane.err || { echo "$tag 4lane failed"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/${tag}_pmc -- python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 > $O/${tag}_pmc.log 2>&1 || { echo "$tag pmc failed"; tail -5 $O/${tag}_pmc.log; exit 1; }
  echo "== $tag"
  python3 - <<PY
import json
for n in ("1lane","4lane"):
    d=json.loads(open("$O/${tag}_"+n+".json").read().strip().splitlines()[-1])
    print(n, d["value"], {k: round(v,4) for k,v in d["stage_ms_per_launch"].items()})
PY
  python3 $R/tools/pmc_summary.py $O/${tag}_pmc
done
