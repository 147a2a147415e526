#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
echo "=== tests P8=1"; DIAMOND_WS_P8=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "conv2d or head or denoiser" > $O/tests_p8.log 2>&1; tail -3 $O/tests_p8.log; grep FAILED $O/tests_p8.log | head
echo "=== conv_bench P8=0"; timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_base.log
echo "=== conv_bench P8=1"; DIAMOND_WS_P8=1 timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_p8.log
for j in 0 1 0 1; do DIAMOND_WS_P8=$j timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_p$j.json 2> $O/bench_p$j.err; python -c "
import json; d=json.load(open('$O/bench_p$j.json')); print('P8=$j bench', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"; done
