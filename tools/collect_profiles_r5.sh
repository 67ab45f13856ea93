#!/bin/bash
# GPU box (through gpurun): the driver's command and its profiles (round 5).  Outputs under gpurun_out/prof_r5/; summaries are made
# by tools/make_profile_summary_r5.py <tag> gpurun_out/prof_r5 and copied to profiles/.
#   1. bench.py default (BASELINE configs[2], mask on, with the mask-off leg)            -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (fewer steps)                 -> trace/
#   3. PMC passes (counter collection serialises kernels: mask-off config, the HIP kernels of the roofline):
#      FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS | SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r5
rm -rf $O; mkdir -p $O
timeout -k 10 500 python3 $R/bench.py --gpus 1 --check > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 2 --leg-steps 20 --cpu-frames 0 --latency-frames 0 > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/bench.py --gpus 1 --config c2 --steps 3 --warmup 1 --cpu-frames 0 --latency-frames 0 > $O/pmc$i.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -3 $O/pmc$i.log; }
done
tail -c 400 $O/bench.json
# 4. one mask lane alone (where a pass spends its time) and the Winograd kernel's counters on the largest layer
cd $R && tools/one_lane_trace.sh r5 > $O/one_lane.log 2>&1; cp $R/gpurun_out/lane_r5.txt $O/one_lane_mask_pass.txt 2>/dev/null
cd $R && tools/r4_w24_pmc.sh "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" > $O/winograd_pmc.txt 2>&1
tail -3 $O/winograd_pmc.txt
# 5. one mask lane alone at the bench's 64 frames per forward, and the one-frame mask pass (the C++ class's per-frame session)
cd $R && sed 's/lane_r4_64/lane_r5_64/g' tools/r4_lane64.sh > /tmp/r5_lane64.sh && bash /tmp/r5_lane64.sh > $O/one_lane64.log 2>&1
cp $R/gpurun_out/lane_r5_64.txt $O/one_lane_mask_pass_64_frames.txt 2>/dev/null; cp $R/gpurun_out/lane_r5_64_sequence.txt $O/one_lane_launches_64_frames.txt 2>/dev/null
cd $R && bash tools/r5_one_frame_trace.sh r5 > $O/one_frame.log 2>&1; cp $R/gpurun_out/one_frame_r5.txt $O/one_frame_mask_pass.txt 2>/dev/null
tail -2 $O/one_frame.log
