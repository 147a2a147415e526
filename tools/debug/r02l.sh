#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -12 $O/tests.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['roofline']['avg_launch_ms'])"
