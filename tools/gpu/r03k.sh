#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for z in 0 1 0 1; do echo "=== conv_bench ZERO=$z"; CONV_BENCH_ZERO=$z timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done
