#!/bin/bash
# Per-launch durations of the denoiser training step (f2) by kernel and grid (tools/trace_by_shape.py).  ~3 GPU-minutes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-train_shapes}; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/ts && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- \
   python $R/bench.py --config train --steps 6 --warmup 3 > $O/bench.log 2>&1)
tail -1 $O/bench.log | cut -c1-200
k=$(find /tmp/ts -name "*kernel_trace.csv" | head -1)
LAST_MS=${LAST_MS:-25} python tools/trace_by_shape.py $k > $O/by_shape.txt 2>&1; cp $k $O/kt.csv; gzip -f $O/kt.csv
head -c 5000 $O/by_shape.txt
