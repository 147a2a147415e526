#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for l in "" ws32 ws128 ws2 ""; do echo "=== lib ${l:-default}"; if [ -n "$l" ]; then export DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_$l.so; else unset DIAMOND_LIB; fi; timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done
