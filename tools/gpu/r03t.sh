#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_proj_fuse.py -q 2>&1 | tail -8
DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_wstrace.so timeout 200 python tools/ws_trace.py 64 2 2>&1 | grep -v amdgpu.ids | tail -6
DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_wstrace.so timeout 200 python tools/ws_trace.py 64 0 2>&1 | grep -v amdgpu.ids | tail -1
for f in 0 1 1; do echo "== bench FUSE=$f"; DIAMOND_FUSE_PROJ=$f timeout 300 python bench.py --steps 2 --warmup 1 2>$O/bench_$f.err | tee $O/bench_$f.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['avg_launch_ms'], {k:v for k,v in list(r['launch_time_ms'].items())[:4]})"; done
timeout 900 python -m pytest tests/test_gpu_tpw.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
