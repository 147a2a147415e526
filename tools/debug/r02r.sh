#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
L=$R/diamond_amd/ablate/libdiamond_hip_norev.so
echo "=== tests REV"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log; grep FAILED $O/tests.log | head
echo "=== conv_bench REV"; timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_rev.log
echo "=== conv_bench no REV"; DIAMOND_LIB=$L timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_norev.log
for v in rev norev rev norev; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = norev ] && lib=$L
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['avg_launch_ms'])"; done
