#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03u; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_proj_fuse.py -q 2>&1 | tail -4
for f in 1 1; do echo "== bench FUSE=$f"; DIAMOND_FUSE_PROJ=$f timeout 300 python bench.py --steps 2 --warmup 1 2>$O/bench_$f.err | tee $O/bench_$f.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['avg_launch_ms'], {k:v for k,v in list(r['launch_time_ms'].items())[:4]})"; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
