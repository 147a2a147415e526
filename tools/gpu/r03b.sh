#!/bin/bash
# Round-3 session B: raw barrier (stores stay in flight) vs __syncthreads; timeline trace; ablations.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
export TMPDIR=/tmp
A=$R/diamond_amd/ablate
echo "=== conv tests (new)"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py tests/test_gpu_precision.py -m gpu -q -x -p no:cacheprovider > $O/tests_conv.log 2>&1; tail -2 $O/tests_conv.log
for v in r02 new sync new r02; do case $v in r02) lib=$A/libdiamond_hip_r02.so;; sync) lib=$A/libdiamond_hip_wssync.so;; *) lib=$R/diamond_amd/libdiamond_hip.so;; esac
  echo "=== conv_bench $v"; DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -6 | tee -a $O/conv_bench_$v.log; done
for n in 162 2 128 16; do echo "=== conv_bench ABL $n"; DIAMOND_LIB=$A/libdiamond_hip_ws$n.so timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $O/conv_bench_abl$n.log; done
echo "=== trace cin64 res0"; DIAMOND_LIB=$A/libdiamond_hip_wstrace.so timeout 120 python tools/ws_trace.py 64 0 2>&1 | grep -v amdgpu.ids | tee $O/trace_64_0.log | tail -20
echo "=== trace cin64 res1"; DIAMOND_LIB=$A/libdiamond_hip_wstrace.so timeout 120 python tools/ws_trace.py 64 1 2>&1 | grep -v amdgpu.ids | tee $O/trace_64_1.log | tail -3
for v in new r02; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$A/libdiamond_hip_r02.so
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['avg_launch_ms'])"; done
