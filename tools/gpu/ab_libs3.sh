#!/bin/bash
# Window / training-step A/B of several library builds on one box (DIAMOND_LIB), alternating twice.
#   bash tools/gpu/ab_libs3.sh <tag> libA.so libB.so [libC.so]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ablibs}; mkdir -p $O; shift
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    echo "== window DIAMOND_LIB=$lib" | tee -a $O/ab_libs.txt
    DIAMOND_LIB=$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(round(d['value'], 1), 'frames/s;', r['kernel'], round(1e3 * r['avg_launch_ms'], 1), 'us;', {k: v for k, v in list(r['launch_time_ms'].items())[:9]})" | tee -a $O/ab_libs.txt
  done
done
for lib in "$@"; do
  echo "== train DIAMOND_LIB=$lib" | tee -a $O/ab_libs.txt
  DIAMOND_LIB=$lib timeout 300 python bench.py --config train --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 3), 'ms/step (graphed);', round(d['eager_ms_per_step'], 3), 'eager; loss', d['loss'])" | tee -a $O/ab_libs.txt
done
