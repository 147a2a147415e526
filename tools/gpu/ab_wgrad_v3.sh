#!/bin/bash
# The weight gradient's third version (prologue as a template parameter, item geometry precomputed) against the second
# (diamond_amd/ablate/libdiamond_hip_wgv2.so = HEAD~'s dmd_backward.hip) on one box: GPU tests, per-shape times, training step, window.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ab_wgrad_v3}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
{
for rep in 1 2; do
  for lib in diamond_amd/libdiamond_hip.so diamond_amd/ablate/libdiamond_hip_wgv2.so; do
    echo "== wgrad_bench DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python tools/wgrad_bench.py wgrad 30 2>&1 | grep -v amdgpu.ids
    echo "== train DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python bench.py --config train --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 3), 'ms/step (graphed);', round(d['eager_ms_per_step'], 3), 'eager; loss', d['loss'])"
    echo "== window DIAMOND_LIB=$lib"
    DIAMOND_LIB=$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-also --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  window', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms')"
  done
done
} 2>&1 | tee $O/ab.txt
