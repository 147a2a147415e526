#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02k2; mkdir -p $O; cd $R
for v in 0 34 162; do echo "=== trace ABL $v"; DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_trace$v.so timeout 120 python tools/debug/ws_trace.py 64 0 2>&1 | grep -E "launch|MEAN|span" | tee -a $O/trace.log; done
