#!/bin/bash
# Same-box A/B of the producer/consumer weight-gradient kernel (DIAMOND_WGRAD_PS=0: the single-role kernel): GPU tests, the
# headline window and the denoiser training step, alternating.  ~6 GPU-minutes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ab_wgrad_ps}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do
  for ps in 1 0; do
    echo "== DIAMOND_WGRAD_PS=$ps"
    DIAMOND_WGRAD_PS=$ps timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  window', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms')"
    DIAMOND_WGRAD_PS=$ps timeout 300 python bench.py --config train --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  train', round(d['value'], 3), 'ms/step; eager', d.get('eager_ms_per_step'), [(k['kernel'][:40], k['avg_us'], k['share_of_launch_time']) for k in d['roofline']['kernels'][:4]])"
  done
done 2>&1 | tee $O/ab.txt
