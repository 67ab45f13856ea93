#!/bin/bash
# Build-container side: timing-experiment variants of the Winograd kernel (results are wrong with any of them) as extra libraries
# under amos-slam_amd/csrc/build/; GPU side: tools/winograd_probe.py --big-only with AMOS_FRONTEND_LIB pointing at each.
set -e
cd "$(dirname "$0")/../amos-slam_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(ls build/amos_*.o | grep -v amos_winograd)
for v in "$@"; do
  case $v in
    M*G*) m=${v#M}; DEF="-DAMOS_WINO_MAP=${m%%G*} -DAMOS_WINO_GROUP=${v#*G}";;   # M1G4: work-group -> (m block, n tile) map 1 with groups of 4
    G*) DEF="-DAMOS_WINO_GROUP=${v#G}";;   # G1, G4, ...: work-groups of an XCD that share a cout tile back to back
    *) DEF="-DAMOS_WINO_EXP_$v";;
  esac
  /opt/rocm/bin/hipcc $FLAGS $DEF -c -o build/wino_$v.o amos_winograd.hip
  /opt/rocm/bin/hipcc $FLAGS -shared -o build/libamos_frontend_$v.so $OBJS build/wino_$v.o 2>/dev/null
  echo built $v
done
