#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
for v in base 2 32 128 34 162 base; do
  lib=$R/diamond_amd/ablate/libdiamond_hip_ws$v.so; [ $v = base ] && lib=$R/diamond_amd/libdiamond_hip.so
  echo "=== ABL $v"; DIAMOND_LIB=$lib timeout 120 python tools/conv_bench.py 2>&1 | grep -E "64x64" | tee -a $O/abl_$v.log
done
