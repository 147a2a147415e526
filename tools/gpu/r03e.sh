#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
A=$R/diamond_amd/ablate
echo "=== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_all.log 2>&1; tail -2 $O/tests_all.log; grep -E "FAILED|ERROR" $O/tests_all.log | head
for v in new ip new ip; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = ip ] && lib=$A/libdiamond_hip_wsip.so
  echo "=== conv_bench $v"; DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done
