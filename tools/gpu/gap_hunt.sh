#!/bin/bash
# What does the host do around a given GPU idle gap?  Kernel trace + HIP API trace + memory-copy trace of one timed steady-state
# window, joined by correlation id (tools/gap_hunt.py).  ~3 GPU-minutes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-gaphunt}; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/gh && timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d /tmp/gh -o t -- \
   python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also > $O/bench.log 2>&1)
tail -1 $O/bench.log | cut -c1-300
find /tmp/gh -name "*.csv" | xargs ls -la
k=$(find /tmp/gh -name "*kernel_trace.csv" | head -1)
a=$(find /tmp/gh -name "*hip_api_trace.csv" | head -1)
m=$(find /tmp/gh -name "*memory_copy_trace.csv" | head -1)
python tools/gap_hunt.py "$k" "$a" "$m" > $O/gap_hunt.txt 2>&1
TRACE_GAPS_DUMP=150 python tools/trace_gaps.py $k 300 330 > $O/gaps.txt 2>&1
head -c 6000 $O/gap_hunt.txt
