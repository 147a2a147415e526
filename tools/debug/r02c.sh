#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
for v in base st1 st2 st3 st4 st8 st16; do
  lib=$R/diamond_amd/ablate/libdiamond_hip_$v.so; [ $v = base ] && lib=$R/diamond_amd/libdiamond_hip.so
  echo "=== $v"; DIAMOND_LIB=$lib timeout 120 python tools/debug/r02b_debug.py stats 2>&1 | grep -E "image pairs differ" | tee -a $O/stats_$v.log
done
