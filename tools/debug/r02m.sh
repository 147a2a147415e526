#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
for v in base 256 128 base; do
  lib=$R/diamond_amd/ablate/libdiamond_hip_ws$v.so; [ $v = base ] && lib=$R/diamond_amd/libdiamond_hip.so
  echo "=== ABL $v"; DIAMOND_LIB=$lib timeout 120 python tools/conv_bench.py 2>&1 | grep -E "64x64|32x32" | tee -a $O/abl_$v.log
done
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "actor_critic or window or linear" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['roofline']['traffic'])"
