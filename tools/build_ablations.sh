#!/bin/bash
# Development aid: builds gpurun_out-independent ablation variants of conv_f16s into diamond_amd/ablate/libdiamond_hip_ablN.so
set -euo pipefail
cd "$(dirname "$0")/../diamond_amd/csrc"
mkdir -p ../ablate build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
for n in "$@"; do
  hipcc $FLAGS -DF16S_ABL=$n -x hip -c dmd_conv_f16.hip -o /tmp/f16_abl$n.o
  objs=$(ls build/*.o | grep -v dmd_conv_f16.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/f16_abl$n.o -o ../ablate/libdiamond_hip_abl$n.so
done
ls -la ../ablate
