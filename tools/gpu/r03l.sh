#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_proj_fuse.py -x -q 2>&1 | tail -15 > $O/tests_proj.log; cat $O/tests_proj.log | tail -8
for f in 0 1 0 1; do echo "== bench FUSE=$f"; DIAMOND_FUSE_PROJ=$f timeout 300 python bench.py --steps 2 --warmup 1 2>$O/bench_$f.err | tee $O/bench_$f.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['avg_launch_ms'], {k:v for k,v in list(r['launch_time_ms'].items())[:5]})"; done
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_tpw.py -x -q 2>&1 | tail -5 > $O/tests_models.log; cat $O/tests_models.log
