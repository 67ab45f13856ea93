#!/bin/bash
# GPU box: duration (GRBM_GUI_ACTIVE / 8 = shader cycles, kernels serialised by the counter pass) and instruction counts of k_fast_cells
# for the phase-removal variants built by tools/orb_variants.sh (e1 .. e5, AMOS_FAST_EXP) and the in-tree library (base).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = base ]; then unset AMOS_FRONTEND_LIB; else export AMOS_FRONTEND_LIB=$R/amos-slam_amd/csrc/build/libamos_frontend_$v.so; fi
  O=$R/gpurun_out/fast_parts/$v
  rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O -- python3 $R/bench.py --config c2 --streams 1 --batch 128 --steps 3 --warmup 1 --cpu-frames 0 > $O.log 2>&1 || { echo "$v failed"; tail -3 $O.log; continue; }
  echo "$v $(python3 $R/tools/pmc_summary.py $O | grep -i fast)"
done
