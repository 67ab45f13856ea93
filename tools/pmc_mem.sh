#!/bin/bash
# GPU box: memory-side counters of the streaming kernels (import, resize levels, blur), one lane alone.
# Each --pmc set is one pass; the four TA counters exceed what the hardware collects together (rocprofv3 aborts with
# "Request exceeds the capabilities of the hardware to collect", error 38): they go two per pass.  A failed pass is
# reported and skipped, the others still run.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/mem
rm -rf $O; mkdir -p $O
ARGS="--config c2 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 --streams 1"
i=0
for set in "TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE TD_TD_BUSY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $ARGS > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -5 $O/p$i.log; continue; }
  echo "== $set"
  python3 $R/tools/pmc_levels.py $O/p$i k_pyramid_level0 1
  python3 $R/tools/pmc_levels.py $O/p$i 'k_pyramid_level<' 7
  python3 $R/tools/pmc_levels.py $O/p$i k_blur 1
done
