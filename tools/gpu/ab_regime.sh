#!/bin/bash
# Same-box A/B of a window WITH episode ends (bench.py --end-rate P, lockstep start) under two settings of environment variables,
# alternating:  gpurun --timeout 400 -- 'bash tools/gpu/ab_regime.sh r05u 0.01 "DIAMOND_WGRAD_DEFER=0 DIAMOND_GN_BWD_FUSED=0" ""'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
TAG=$1; P=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for s in "$@"; do
    echo "== end-rate $P [$s]"
    env $s timeout 200 python bench.py --end-rate $P --steps ${STEPS:-5} --warmup 3 --no-also --no-cpu-baseline --no-exact-fp32 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'frames/s', d['step_ms'], d['config']['env_stats'])"
  done
done 2>&1 | tee $O/ab_regime_$P.txt
