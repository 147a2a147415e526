#!/bin/bash
# Where is the GPU idle inside a window?  rocprofv3 kernel traces of the headline window (nobody ends), of the steady state of
# the reference's training loop (episode lengths spread over the horizon + ends at p = 0.003 per step) and of the window with the
# unbiased synthetic reward / end model (half of the envs end at every step), reduced with tools/trace_gaps.py.  ~4 GPU-minutes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-gaps}; mkdir -p $O
export TMPDIR=/tmp
for mode in no_ends steady_p003 unbiased; do
  extra="--no-ends"; [ $mode = unbiased ] && extra="--no-end-logit-bias"; [ $mode = steady_p003 ] && extra=""  # (the steady state is bench.py's default since round 6)
  (cd /tmp && rm -rf /tmp/gaps_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gaps_$mode -o t -- \
     python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also $extra > $O/$mode.log 2>&1)
  t=$(find /tmp/gaps_$mode -name "*kernel_trace.csv" | head -1)
  k=$(find /tmp/gaps_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$k" ] && cp $k $O/${mode}_kernel_stats.csv
  [ -n "$t" ] && python tools/trace_gaps.py $t 300 400 > $O/${mode}_gaps.txt 2>&1
  tail -1 $O/$mode.log | cut -c1-200; head -14 $O/${mode}_gaps.txt
done
