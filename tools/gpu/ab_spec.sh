#!/bin/bash
# same-box A/B of the speculative sampler step (env_loop / WorldModelEnv.step_end_issue): policy-only speculation vs both
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/ab_spec; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -x -k "full_window or speculative or teacher" 2>&1 | tail -3 | tee $O/tests.txt
STEPS=3 bash tools/gpu/ab_bench.sh DIAMOND_SPECULATIVE_POLICY policy 1 2>&1 | tee $O/ab.txt
for m in policy 1; do
  DIAMOND_SPECULATIVE_POLICY=$m timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also --no-end-logit-bias 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unbiased, speculation=$m', round(d['value'], 1), 'frames/s')" | tee -a $O/ab.txt
done
