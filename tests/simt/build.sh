#!/bin/bash
# TEST INFRASTRUCTURE: diamond_amd/csrc/*.hip compiled as HOST C++ against the SIMT interpreter (simt.h) ->
# tests/simt/_build/libdiamond_simt.so, loaded by tests/test_simt_kernels.py only.
set -euo pipefail
cd "$(dirname "$0")"
CXX=${SIMT_CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=../../diamond_amd/csrc
FLAGS="-x c++ -std=c++17 -O2 -g0 -fPIC -mf16c -mfma -mavx2 -ffp-contract=off -Iinclude -Wno-unused-value -Wno-unknown-attributes -Wno-unused-result -Wno-psabi"
mkdir -p _build
pids=()
for f in dmd_conv.hip dmd_conv1x1.hip dmd_conv_f16ws.hip dmd_backward.hip dmd_linear.hip dmd_attention.hip dmd_pointwise.hip dmd_lowres.hip dmd_pack.hip dmd_capi.cpp; do
  o=_build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ] || [ $SRC/dmd_common.h -nt "$o" ] || [ ../../include/diamond_hip.h -nt "$o" ] || [ simt.h -nt "$o" ] || [ include/hip/hip_runtime.h -nt "$o" ]; then
    ( $CXX $FLAGS -c "$SRC/$f" -o "$o" ) &
    pids+=($!)
  fi
done
for f in simt.cpp stubs.cpp; do
  o=_build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ simt.h -nt "$o" ] || [ include/hip/hip_runtime.h -nt "$o" ]; then
    ( $CXX $FLAGS -O2 -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc -eq 0 ] || { echo "simt build failed"; exit 1; }
# link only when an object is newer than the library, into a scratch name that is then renamed over it: a process that has the
# library mapped keeps its (old) file, one that is about to dlopen it never finds it missing or half written
OUT=_build/libdiamond_simt.so
need=0
[ -f "$OUT" ] || need=1
for o in _build/*.o; do [ "$o" -nt "$OUT" ] && need=1; done
if [ $need -eq 1 ]; then
  $CXX -shared -fPIC _build/*.o -o "$OUT.$$.tmp" -lpthread -lm
  mv -f "$OUT.$$.tmp" "$OUT"
fi
echo "built $(realpath $OUT)"
