#!/bin/bash
# A/B: previous build (prev) vs current (reissue after phase 1, residual prefetch)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
A=$R/diamond_amd/ablate
echo "=== conv tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py tests/test_gpu_precision.py -m gpu -q -x -p no:cacheprovider > $O/tests_conv.log 2>&1; tail -2 $O/tests_conv.log
for v in prev new prev new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = prev ] && lib=$A/libdiamond_hip_prev.so
  echo "=== conv_bench $v"; DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -6; done
for v in prev new prev new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = prev ] && lib=$A/libdiamond_hip_prev.so
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', d['value'])"; done
