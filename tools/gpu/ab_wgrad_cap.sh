#!/bin/bash
# Same-box A/B of the weight-gradient launch plan's workgroup cap (DIAMOND_WGRAD_MAX_WG): 256 = one workgroup per CU (round 4's choice,
# measured on the 64-channel instance whose 105 KB of LDS allow one per CU anyway); the 16 / 32-channel instances of the actor-critic
# encoder use 64 KB: two fit a CU.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ab_wgrad}; mkdir -p $O
for rep in 1 2; do
  for cap in 256 512 768 1024; do
    DIAMOND_WGRAD_MAX_WG=$cap timeout 300 python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lt = d['roofline']['launch_time_ms']
print('cap=$cap', round(d['value'], 1), 'frames/s', {k: v for k, v in lt.items() if k.startswith('wgrad') or 'wgrad' in k})"
  done
done 2>&1 | tee $O/ab_wgrad_cap.txt
