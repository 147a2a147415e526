#!/bin/bash
# Same-box A/B of the denoiser training step (bench.py --config train: f2) under settings of environment variables, alternating:
#   gpurun --timeout 900 -- 'bash tools/gpu/ab_train.sh r05t "DIAMOND_WGRAD_DEFER=0 DIAMOND_TRAIN_FUSE_PROJ=0" ""'
# (each argument after the tag: one setting = a space-separated list of VAR=value, "" = the defaults), then rocprofv3 kernel stats
# of the default setting (kernels per replayed step).  Boxes of the pool differ by several per cent: compare within one call only.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
  for s in "$@"; do
    for b in ${BATCHES:-32 256}; do
      echo "== [$s] batch $b"
      env $s timeout 300 python bench.py --config train --batch $b --steps ${STEPS:-40} --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 3), 'ms/step;', round(d['frames_per_s'], 1), 'frames/s; eager', d['eager_ms_per_step'] and round(d['eager_ms_per_step'], 2), 'ms; loss', d['loss'])"
    done
  done
done 2>&1 | tee $O/ab_train.txt
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o train -- \
   python $R/bench.py --config train --steps 40 --warmup 3 > $O/prof_train.log 2>&1; echo "rocprof rc=$?"
 f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/train_kernel_stats.csv && head -8 $f | cut -c1-150)
