#!/bin/bash
# Build-container side: variants of the ORB kernels (amos_orb.hip compiled with extra -D flags) as extra libraries under
# amos-slam_amd/csrc/build/libamos_frontend_<name>.so; GPU side: AMOS_FRONTEND_LIB=<that file> python bench.py --config c2 ...
# usage: tools/orb_variants.sh name1="-DAMOS_X=1 -DAMOS_Y=2" name2="..."
set -e
cd "$(dirname "$0")/../amos-slam_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(ls build/amos_*.o | grep -v amos_orb)
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/orb_$name.o amos_orb.hip
  /opt/rocm/bin/hipcc $FLAGS -shared -o build/libamos_frontend_$name.so $OBJS build/orb_$name.o 2>/dev/null
  echo built $name "($defs)"
done
