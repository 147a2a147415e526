#!/bin/bash
# Round-3 session G: full GPU suite (new tests: RCCL world 1, heun5, 256x256 B=8, tightened budgets), bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
export TMPDIR=/tmp
A=$R/diamond_amd/ablate
echo "=== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log; grep -E "^FAILED|^ERROR" $O/tests_all.log | head -20
grep -h "quantised frame\|heun5 step\|256x256 B=8\|forced pooling" $O/tests_all.log | head -40
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for v in new r02; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$A/libdiamond_hip_r02.so
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); print('cfg4', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac']); print({k:v for k,v in list(d['roofline']['launch_time_ms'].items())[:8]})"
