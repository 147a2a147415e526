#!/bin/bash
# Same-box A/B of bench.py (configs[1]) under two settings of one environment variable, alternating, e.g.
#   gpurun --timeout 1500 -- 'bash tools/gpu/ab_bench.sh DIAMOND_LOWRES_CHAIN 0 3'
#   gpurun --timeout 1500 -- 'bash tools/gpu/ab_bench.sh DIAMOND_LIB diamond_amd/ablate/libdiamond_hip_r02.so diamond_amd/libdiamond_hip.so'
# Boxes of the pool differ by several per cent at identical code: only numbers of one call are comparable.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
VAR=$1; A=$2; B=$3
for v in "$A" "$B" "$A" "$B"; do
  echo "== $VAR=$v"
  env $VAR="$v" timeout 300 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(round(d['value'], 1), 'frames/s;', r['kernel'], round(1e3 * r['avg_launch_ms'], 1), 'us;', {k: v for k, v in list(r['launch_time_ms'].items())[:5]})"
done
