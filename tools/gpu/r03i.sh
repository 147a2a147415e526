#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
A=$R/diamond_amd/ablate
for n in 0 4 2 160 166; do echo "=== trace ABL $n (cin64 res0)"; DIAMOND_LIB=$A/libdiamond_hip_wstr$n.so timeout 120 python tools/ws_trace.py 64 0 2>&1 | grep -v amdgpu.ids | tee $O/trace_abl$n.log | grep "launch\|MEAN"; done
