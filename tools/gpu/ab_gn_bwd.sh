#!/bin/bash
# GroupNorm backward with GN_BWD_U requests per thread and trip against the library before it (diamond_amd/ablate/libdiamond_hip_head.so):
# the GroupNorm / training GPU tests, the training step and the window, alternating; the per-kernel table of the training step.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ab_gn_bwd}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
{
for rep in 1 2 3; do
  for lib in diamond_amd/libdiamond_hip.so diamond_amd/ablate/libdiamond_hip_head.so; do
    echo "== train DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python bench.py --config train --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 3), 'ms/step (graphed);', [(k['kernel'][:28], k['calls'], k['avg_us']) for k in d['roofline']['kernels'][:3]])"
    echo "== window DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  window', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms; gn_silu_bwd ms', d['roofline']['launch_time_ms'].get('dmd_gn_silu_bwd'))"
  done
done
} 2>&1 | tee $O/ab.txt
