#!/bin/bash
# The FiLM tables of a frame's denoising steps computed together (default) against per step (DIAMOND_BATCH_FILM=0): GPU suite, then the
# window, the B = 1 latency line and configs[4], alternating on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ab_film}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
{
for rep in 1 2 3; do
  for f in 1 0; do
    echo "== DIAMOND_BATCH_FILM=$f"
    DIAMOND_BATCH_FILM=$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lt = d['roofline']['launch_time_ms']
print('  window', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms; linear ms', round(lt.get('linear_mfma_kernel<true>', 0) + lt.get('linear_mfma_kernel<false>', 0), 2))"
    DIAMOND_BATCH_FILM=$f timeout 300 python bench.py --config latency --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  latency', round(d['value'], 4), d['unit'])"
  done
done
for f in 1 0 1 0; do
  echo "== configs[4] DIAMOND_BATCH_FILM=$f"
  DIAMOND_BATCH_FILM=$f timeout 300 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cfg4', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms')"
done
} 2>&1 | tee $O/ab.txt
