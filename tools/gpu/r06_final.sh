#!/bin/bash
# Round 6 evidence set: tools/gpu/final_round.sh (GPU suite, smoke, the driver's bench invocation, rocprofv3 kernel stats of a bench run,
# PMC traffic of a bench window) + the PMC passes of configs[4] + the idle-gap trace of the headline (steady-state) window.
set -u
TAG=${1:-r06final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu/final_round.sh $TAG
CFG=4 bash tools/pmc_collect.sh > $O/pmc_cfg4.log 2>&1; echo "pmc cfg4 rc=$?"
mkdir -p $O/pmc_cfg4 && cp gpurun_out/pmc_cfg4/*.json $O/pmc_cfg4/ 2>/dev/null
for mode in steady_p003 no_ends; do
  extra=""; [ $mode = no_ends ] && extra="--no-ends"
  (cd /tmp && rm -rf /tmp/gaps_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gaps_$mode -o t -- \
     python $R/bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also $extra > $O/$mode.log 2>&1)
  t=$(find /tmp/gaps_$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/trace_gaps.py $t 300 600 > $O/${mode}_gaps.txt 2>&1
  grep '"metric"' $O/$mode.log | cut -c1-140; head -2 $O/${mode}_gaps.txt
done
