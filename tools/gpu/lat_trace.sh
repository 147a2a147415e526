#!/bin/bash
# B = 1 timeline of conv_f16ws (trace build) + wgrad v3 (forced 128 registers for 32 -> 32) against v2
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-lat_trace}; mkdir -p $O
export TMPDIR=/tmp
{
for args in "64 1 64 64 1" "64 1 64 16 1" "128 0 64 64 1" "64 2 64 64 1" "32 1 32 64 1"; do
  echo "== ws_trace $args"
  DIAMOND_LIB=diamond_amd/ablate/libdiamond_hip_wstrace.so timeout 120 python tools/ws_trace.py $args 2>&1 | grep -v amdgpu.ids
done
} > $O/trace.txt 2>&1
{
for rep in 1 2; do
  for lib in diamond_amd/libdiamond_hip.so diamond_amd/ablate/libdiamond_hip_wgv2.so; do
    echo "== wgrad_bench DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python tools/wgrad_bench.py wgrad 30 ac 2>&1 | grep -v amdgpu.ids
  done
done
} 2>&1 | tee $O/ab_wgrad_v3b.txt
