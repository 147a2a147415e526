#!/bin/bash
# Development aid (DMD_LAB): timing-proxy variants of attention_f16x2_kernel (WRONG results) into diamond_amd/ablate/libdiamond_hip_afN.so
#   bash tools/build_attention_ablations.sh 1 2 4 8 16 32 64   (AF_ABL bits, see dmd_attention.hip)
set -euo pipefail
cd "$(dirname "$0")/../diamond_amd/csrc"
mkdir -p ../ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
for n in "$@"; do
  hipcc $FLAGS -DDMD_LAB -DAF_ABL=$n -x hip -c dmd_attention.hip -o /tmp/att_abl$n.o
  objs=$(ls build/*.o | grep -v dmd_attention.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/att_abl$n.o -o ../ablate/libdiamond_hip_af$n.so
done
