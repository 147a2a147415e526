#!/bin/bash
# Round 6: the slots env loop on the GPU -- parity suite, same-process A/B against the round-5 pipelined loop and the sequential
# order in every regime, the idle-gap trace of the steady-state window, the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-r06b}; mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 -x > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
fi
timeout 600 python tools/ab_env_loop.py > $O/ab_env_loop.json 2> $O/ab_env_loop.txt; echo "ab rc=$?"; grep "^\[1\]" $O/ab_env_loop.txt | cut -c1-150
for mode in steady_p003 no_ends; do
  extra=""; [ $mode = no_ends ] && extra="--no-ends"
  (cd /tmp && rm -rf /tmp/gaps_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gaps_$mode -o t -- \
     python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also $extra > $O/$mode.log 2>&1)
  t=$(find /tmp/gaps_$mode -name "*kernel_trace.csv" | head -1)
  k=$(find /tmp/gaps_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$k" ] && cp $k $O/${mode}_kernel_stats.csv
  [ -n "$t" ] && python tools/trace_gaps.py $t 300 400 > $O/${mode}_gaps.txt 2>&1
  grep '"metric"' $O/$mode.log | cut -c1-160; head -2 $O/${mode}_gaps.txt
done
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_line.json; tail -3 $O/bench.err
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
