#!/bin/bash
# Same-box A/B of several library builds (DIAMOND_LIB) over the bench window, alternating twice:
#   bash tools/gpu/ab_libs_bench.sh <tag> lib1.so lib2.so ...      (the shipping library is always the first arm)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ablibs}; mkdir -p $O; shift
for rep in 1 2; do
  for lib in diamond_amd/libdiamond_hip.so "$@"; do
    DIAMOND_LIB=$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$lib', round(d['value'], 1), 'frames/s;', {k.replace('conv_f16ws_kernel', 'ws'): v for k, v in list(r['launch_time_ms'].items())[:4]})"
  done
done 2>&1 | tee $O/ab_libs_bench.txt
