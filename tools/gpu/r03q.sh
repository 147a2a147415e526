#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_wstrace.so
for r in 0 1 2; do echo "=== trace res=$r"; timeout 200 python tools/ws_trace.py 64 $r 2>&1 | grep -v amdgpu.ids | tail -8; done
