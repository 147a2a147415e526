#!/bin/bash
# Round-4 same-box A/Bs after the staged kernels were adopted (wgrad: prefetching 32-pixel kernel, 256 workgroups) or deleted:
# what is left to decide -- GroupNorm backward's folded channel sums, and the old 16-pixel wgrad kernel on the headline window.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/ab4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pack_jobs.py tests/test_gpu_train_graph.py -q -p no:cacheprovider -x -k "wgrad or gn_silu or pack or copies or graph or swapped" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
run() { local label=$1; shift
  env "$@" timeout 120 python bench.py --config train 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'graphed ms/step', round(d['ms_per_step'], 3), 'eager', round(d.get('eager_ms_per_step', 0), 3))" | tee -a $O/train.txt; }
: > $O/train.txt
for rep in 1 2; do
  run "default(mode3,wg256)" X=1
  run "gnfold" DIAMOND_GN_BWD_FOLD=1
  run "mode1/wg256" DIAMOND_WGRAD_MODE=1
done
: > $O/window.txt
for rep in 1 2; do
  for setting in "X=1" "DIAMOND_WGRAD_MODE=1 DIAMOND_WGRAD_MAX_WG=1024" "DIAMOND_GN_BWD_FOLD=1"; do
    env $setting timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['launch_time_ms']
print('configs[1]', '$setting', round(d['value'], 1), 'frames/s; wgrad', r.get('dmd_conv2d_wgrad'), 'gn_bwd', r.get('dmd_gn_silu_bwd'))" | tee -a $O/window.txt
  done
done
