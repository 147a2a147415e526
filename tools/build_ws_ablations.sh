#!/bin/bash
# Development aid: ablation variants of conv_f16ws into diamond_amd/ablate/libdiamond_hip_wsN.so
set -euo pipefail
cd "$(dirname "$0")/../diamond_amd/csrc"
mkdir -p ../ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
for n in "$@"; do
  hipcc $FLAGS ${WS_EXTRA:-} -DWS_ABL=$n -x hip -c dmd_conv_f16ws.hip -o /tmp/f16ws_abl$n.o
  objs=$(ls build/*.o | grep -v dmd_conv_f16ws.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/f16ws_abl$n.o -o ../ablate/libdiamond_hip_ws$n.so
done
