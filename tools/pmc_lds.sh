#!/bin/bash
# GPU box: LDS-side counters of the extractor kernels (one lane of 128 frames alone on the chip).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lds
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; continue; }
  python3 $R/tools/pmc_summary.py $O/p$i
done
