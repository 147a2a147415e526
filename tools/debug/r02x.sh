#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02x; mkdir -p $O; cd $R
L=$R/diamond_amd/ablate/libdiamond_hip_adma.so
echo "=== tests A_DMA"; DIAMOND_LIB=$L timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "conv2d or head or denoiser" > $O/tests.log 2>&1; tail -2 $O/tests.log; grep FAILED $O/tests.log | head -12
echo "=== conv_bench base"; timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_base.log
echo "=== conv_bench A_DMA"; DIAMOND_LIB=$L timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_adma.log
for v in base adma base adma; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = adma ] && lib=$L
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['avg_launch_ms'])"; done
