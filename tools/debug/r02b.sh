#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 600 python tools/debug/r02b_debug.py > $O/debug.log 2>&1; echo "debug rc=$?"; tail -40 $O/debug.log
echo "=== conv_bench baseline (WS_PIPE=0)"; timeout 200 python tools/conv_bench.py 2>&1 | tee $O/conv_bench_base.log
echo "=== conv_bench pipelined"; DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_pipe.so timeout 200 python tools/conv_bench.py 2>&1 | tee $O/conv_bench_pipe.log
echo "=== conv_bench baseline again"; timeout 200 python tools/conv_bench.py 2>&1 | tee $O/conv_bench_base2.log
DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_pipe.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "conv2d or head" > $O/tests_pipe.log 2>&1; tail -3 $O/tests_pipe.log
DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_pipe.so timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_pipe.json 2> $O/bench_pipe.err; python -c "
import json; d=json.load(open('$O/bench_pipe.json')); print('pipe bench', d['value'], d['roofline']['avg_launch_ms'])"
timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_base.json 2> $O/bench_base.err; python -c "
import json; d=json.load(open('$O/bench_base.json')); print('base bench', d['value'], d['roofline']['avg_launch_ms'])"
