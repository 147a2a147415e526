#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
for a in "64 0" "128 0" "64 1"; do echo "=== trace cin/res $a"; DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_trace.so timeout 120 python tools/debug/ws_trace.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.log; done
