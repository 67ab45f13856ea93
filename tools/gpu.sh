#!/bin/bash
# Rebuild every native library, then run a command on the GPU box:  tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
make -s -C amos-slam_amd/csrc
make -s -C amos-slam_amd/host
make -s -C oracle
exec /usr/local/graft/bin/gpurun "$@"
