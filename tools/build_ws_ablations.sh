#!/bin/bash
# Development aid (DMD_LAB): timing-proxy variants of conv_f16ws (WRONG results) into diamond_amd/ablate/libdiamond_hip_wsN.so
#   bash tools/build_ws_ablations.sh 162     (N = WS_ABL bits: 2 no activation loads, 16 no MFMAs, 32 no weight movement, 128 no stores)
#   bash tools/build_ws_ablations.sh trace   (correct results + the s_memtime stamps tools/ws_trace.py reads: -DWS_TRACE)
set -euo pipefail
cd "$(dirname "$0")/../diamond_amd/csrc"
mkdir -p ../ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
for n in "$@"; do
  abl="-DWS_ABL=$n"; [ "$n" = trace ] && abl="-DWS_ABL=0 -DWS_TRACE"
  hipcc $FLAGS ${WS_EXTRA:-} -DDMD_LAB $abl -x hip -c dmd_conv_f16ws.hip -o /tmp/f16ws_abl$n.o
  objs=$(ls build/*.o | grep -v dmd_conv_f16ws.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/f16ws_abl$n.o -o ../ablate/libdiamond_hip_ws$n.so
done
