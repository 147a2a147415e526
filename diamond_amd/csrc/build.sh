#!/bin/bash
# Build libdiamond_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output:
# diamond_amd/libdiamond_hip.so (git-ignored, travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdiamond_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
mkdir -p build
pids=()
for f in dmd_conv.hip dmd_conv1x1.hip dmd_conv_f16ws.hip dmd_backward.hip dmd_linear.hip dmd_attention.hip dmd_pointwise.hip dmd_lowres.hip dmd_pack.hip dmd_capi.cpp; do
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ dmd_common.h -nt "$o" ] || [ ../../include/diamond_hip.h -nt "$o" ] || [ extra_flags.txt -nt "$o" ]; then
    extra=$(grep -v '^#' extra_flags.txt | awk -v f="$f" '$1 == f { $1 = ""; print }')  # per-source flags (extra_flags.txt)
    ( hipcc $FLAGS $extra -x hip -c "$f" -o "$o" ${EXTRA_HIPCC_FLAGS:-} ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o "$OUT"
echo "built $(realpath $OUT)"
