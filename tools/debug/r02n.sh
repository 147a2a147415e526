#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
for c in 0 128 64 0; do DIAMOND_DENOISE_CHUNK=$c timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_c$c.json 2> $O/bench_c$c.err; python -c "
import json; d=json.load(open('$O/bench_c$c.json')); print('chunk $c bench', d['value'], d['roofline']['avg_launch_ms'])"; done
