#!/bin/bash
# Round 6, first GPU call: the whole GPU suite (wide-configuration twins included), SQ counters of the shipped library, the
# ordered kernel list of a steady-state step, and the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-r06a}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 -x > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -5 $O/tests.log
bash tools/pmc_sq.sh r06 > $O/pmc_sq.log 2>&1; cp -r gpurun_out/pmc_sq_r06 $O/ 2>/dev/null; ls $O/pmc_sq_r06
for mode in steady_p003 no_ends; do
  extra=""; [ $mode = steady_p003 ] && extra="--stagger --end-rate 0.003"
  (cd /tmp && rm -rf /tmp/gaps_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gaps_$mode -o t -- \
     python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also $extra > $O/$mode.log 2>&1)
  t=$(find /tmp/gaps_$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && TRACE_GAPS_DUMP=80 python tools/trace_gaps.py $t 300 400 > $O/${mode}_gaps.txt 2>&1
  tail -1 $O/$mode.log | cut -c1-200
done
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench_line.json
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
