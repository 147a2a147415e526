#!/bin/bash
# The staged weight-gradient kernels (DIAMOND_WGRAD_MODE 2 / 3, DIAMOND_WGRAD_MAX_WG; diamond_amd/csrc/dmd_backward.hip) on the
# GPU: parity first, then same-box A/Bs of the denoiser training step and of the headline window.  ~7 GPU-minutes.
#   gpurun --timeout 600 -- 'bash tools/gpu/staged_wgrad.sh'
# Results -> gpurun_out/staged_wgrad/.  Whatever wins becomes the default in launch_wgrad / wgrad_plan; the rest is deleted.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/staged_wgrad; mkdir -p $O
DIAMOND_STAGED_TESTS=1 timeout 320 python -m pytest tests/test_gpu_staged.py -q -p no:cacheprovider -k "wgrad or training_step" > $O/tests.log 2>&1
echo "staged tests rc=$?"; tail -3 $O/tests.log
run() {  # label, env assignments...
  local label=$1; shift
  env "$@" timeout 120 python bench.py --config train 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'graphed ms/step', round(d['ms_per_step'], 3), 'eager', round(d.get('eager_ms_per_step', 0), 3))" | tee -a $O/ab.txt
}
: > $O/ab.txt
for rep in 1 2; do
  run "mode1/wg1024" DIAMOND_WGRAD_MODE=1
  run "mode2/wg1024" DIAMOND_WGRAD_MODE=2
  run "mode3/wg1024" DIAMOND_WGRAD_MODE=3
  run "mode1/wg256 " DIAMOND_WGRAD_MODE=1 DIAMOND_WGRAD_MAX_WG=256
  run "mode3/wg256 " DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=256
  run "mode3/wg512 " DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=512
  run "mode3/wg256/1pass" DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=256 DIAMOND_WGRAD_SINGLE_REDUCE=256
  run "mode3/wg256/1pass/gnfold" DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=256 DIAMOND_WGRAD_SINGLE_REDUCE=256 DIAMOND_GN_BWD_FOLD=1
  # ... and the few-tile convolution for the forward / data-gradient launches of the 16x16 level (32 tiles at batch 32)
  run "all + lat64" DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=256 DIAMOND_WGRAD_SINGLE_REDUCE=256 DIAMOND_GN_BWD_FOLD=1 DIAMOND_CONV_LATENCY_TILES=64
done
# the same switches on the headline window (configs[1]: the actor-critic backward over 256 x 15 frames has weight gradients
# and GroupNorm backward too: 3 % + 1.3 % of the window)
for rep in 1 2; do
  for setting in "DIAMOND_WGRAD_MODE=1" "DIAMOND_WGRAD_MODE=3 DIAMOND_WGRAD_MAX_WG=256 DIAMOND_WGRAD_SINGLE_REDUCE=256 DIAMOND_GN_BWD_FOLD=1"; do
    env $setting timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1]', '$setting', round(d['value'], 1), 'frames/s')" | tee -a $O/ab.txt
  done
done
