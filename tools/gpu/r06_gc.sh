#!/bin/bash
# Is the one ~20 ms hole per window python's cyclic GC?  Same box, alternating: default / gc off / gc per window; then a trace with context.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-r06d}; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for mode in default off window; do
    timeout 300 python bench.py --gc $mode --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-exact-fp32 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc=$mode', round(d['value'], 1), 'frames/s', d['step_ms'])"
  done
done 2>&1 | tee $O/ab_gc.txt
(cd /tmp && rm -rf /tmp/gaps_gc && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_gc -o t -- \
   python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also > $O/trace.log 2>&1)
t=$(find /tmp/gaps_gc -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_gaps.py $t 300 400 > $O/steady_gaps_context.txt 2>&1
grep -A11 "^gap of" $O/steady_gaps_context.txt | head -60
