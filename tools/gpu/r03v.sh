#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tpw.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -6
for p8 in 0 1 0 1; do echo "=== conv_bench c32 P8=$p8"; DIAMOND_WS_P8=$p8 CONV_BENCH_COUT=32 timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done
for p8 in 0 1 1; do echo "== bench P8=$p8"; DIAMOND_WS_P8=$p8 timeout 300 python bench.py --steps 2 --warmup 1 2>$O/bench_$p8.err | tee $O/bench_$p8.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['avg_launch_ms'], {k:v for k,v in list(r['launch_time_ms'].items())[:5]})"; done
