#!/bin/bash
# Build-container side: timing-experiment variants of the F(2 x 4) Winograd kernel (results are wrong with the EXP ones) as extra libraries
# under amos-slam_amd/csrc/build/libamos_frontend_w24_<name>.so; GPU side: AMOS_FRONTEND_LIB=<that> python tools/winograd_probe.py --f24-time
# usage: tools/w24_variants.sh NOX NOU NOT NOBAR NOEPI G1 G8 ...
set -e
cd "$(dirname "$0")/../amos-slam_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(ls build/amos_*.o | grep -v amos_winograd24)
for v in "$@"; do
  case $v in
    ALL) DEF="-DAMOS_W24_EXP_NOX -DAMOS_W24_EXP_NOU -DAMOS_W24_EXP_NOT -DAMOS_W24_EXP_NOEPI ";;           # the MFMA loop alone (with its barriers)
    ALLB) DEF="-DAMOS_W24_EXP_NOX -DAMOS_W24_EXP_NOU -DAMOS_W24_EXP_NOT -DAMOS_W24_EXP_NOEPI -DAMOS_W24_EXP_NOBAR ";;  # ... without barriers
    XU) DEF="-DAMOS_W24_EXP_NOX -DAMOS_W24_EXP_NOU ";;    # no global traffic at all
    G*) DEF="-DAMOS_W24_GROUP=${v#G}";;
    D*) DEF="-D${v#D}";;
    *) DEF="-DAMOS_W24_EXP_$v";;
  esac
  /opt/rocm/bin/hipcc $FLAGS $DEF -c -o build/w24_$v.o amos_winograd24.hip
  /opt/rocm/bin/hipcc $FLAGS -shared -o build/libamos_frontend_w24_$v.so $OBJS build/w24_$v.o 2>/dev/null
  echo built $v
done
