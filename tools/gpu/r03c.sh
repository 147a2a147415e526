#!/bin/bash
# Round-3 session C: producer VALU diet (c, d tables, v_fma_mix split, saddr loads), bias through the accumulators, DPP statistics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
export TMPDIR=/tmp
A=$R/diamond_amd/ablate
echo "=== all gpu tests (new)"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log; grep -E "FAILED|ERROR" $O/tests_all.log | head -20
for v in r02 new r02 new; do case $v in r02) lib=$A/libdiamond_hip_r02.so;; *) lib=$R/diamond_amd/libdiamond_hip.so;; esac
  echo "=== conv_bench $v"; DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_$v.log; done
for n in ; do echo "=== conv_bench ABL $n"; DIAMOND_LIB=$A/libdiamond_hip_ws$n.so timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $O/conv_bench_abl$n.log; done
echo "=== conv_bench cout32"; for v in r02 new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$A/libdiamond_hip_r02.so; CONV_BENCH_COUT=32 DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench32_$v.log; done
echo "=== trace cin64 res0"; DIAMOND_LIB=$A/libdiamond_hip_wstrace.so timeout 120 python tools/ws_trace.py 64 0 2>&1 | grep -v amdgpu.ids | tee $O/trace_64_0.log | tail -12
for v in new r02 new r02; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$A/libdiamond_hip_r02.so
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['avg_launch_ms'])"; done
