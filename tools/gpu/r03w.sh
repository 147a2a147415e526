#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03w; mkdir -p $O
for g in 0 1 0 1; do echo "== bench GRAPH_SAMPLER=$g"; DIAMOND_GRAPH_SAMPLER=$g timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 2>$O/bench_$g.err | tee $O/bench_$g.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['avg_launch_ms'], sum(r['launch_time_ms'].values()))"; tail -2 $O/bench_$g.err | cut -c1-300; done
