#!/bin/bash
# GPU box: issue / LDS counters of the extractor kernels, one lane of 128 frames alone on the chip (bench.py --config c2 --streams 1).
# usage: tools/r4_pmc_fast.sh <out name> [kernel substring to print]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
K=${2:-fast}
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; continue; }
  python3 $R/tools/pmc_summary.py $O/p$i | grep -i "$K"
done
