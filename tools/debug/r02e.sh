#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
echo "=== tests JOINT=1"; DIAMOND_WS_JOINT=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "conv2d or head or denoiser" > $O/tests_joint.log 2>&1; tail -3 $O/tests_joint.log; grep FAILED $O/tests_joint.log | head
echo "=== tests JOINT=0 (goff-free producers)"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "conv2d or head or denoiser" > $O/tests_base.log 2>&1; tail -3 $O/tests_base.log; grep FAILED $O/tests_base.log | head
echo "=== conv_bench JOINT=0"; timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_base.log
echo "=== conv_bench JOINT=1"; DIAMOND_WS_JOINT=1 timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_joint.log
for j in 0 1 0 1; do DIAMOND_WS_JOINT=$j timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_j$j.json 2> $O/bench_j$j.err; python -c "
import json; d=json.load(open('$O/bench_j$j.json')); print('JOINT=$j bench', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"; done
