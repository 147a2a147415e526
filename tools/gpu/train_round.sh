#!/bin/bash
# One call for a change to the training step (f2): the -m gpu suite, the same-box A/B of bench.py --config train (old switches vs
# defaults, batches 32 and 256, alternating), rocprofv3 kernel stats of the default training step, a quick A/B of the headline
# window (the actor-critic backward shares the kernels), and the driver's own bench invocation.  ~9 GPU-minutes.
set -u
TAG=${1:-train}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
OLD="DIAMOND_WGRAD_DEFER=0 DIAMOND_TRAIN_FUSE_PROJ=0 DIAMOND_GN_BWD_FUSED=0"
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
BATCHES="32 256" STEPS=40 bash tools/gpu/ab_train.sh $TAG "$OLD" ""
for s in "$OLD" "" "$OLD" ""; do
  echo "== window [$s]"
  env $s timeout 200 python bench.py --steps 4 --warmup 2 --no-also --no-cpu-baseline --no-exact-fp32 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'frames/s', d['step_ms'])"
done 2>&1 | tee $O/ab_window.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python -c "
import json; d = json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1])
print('headline', round(d['value'], 1), 'train', d['also']['train'], 'latency', d['also'].get('latency'))" 2>&1 | cut -c1-600
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
