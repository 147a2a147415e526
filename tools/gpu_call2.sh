#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite, the env loop A/B under episodes that end (merged encoder pass, adaptive
# speculation), the default bench line, rocprofv3 kernel stats and the PMC traffic passes of the bench window.
set -u
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 -x > $O/tests.log 2>&1
echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
timeout 500 python tools/ab_env_loop.py > $O/ab_env_loop.json 2> $O/ab_env_loop.err; echo "ab rc=$?"; tail -50 $O/ab_env_loop.err
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench_line.json; tail -5 $O/bench.err
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- \
   python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
 f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -8 $f)
(cd /tmp && rm -rf /tmp/profs_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$TAG -o bench -- \
   python $R/bench.py --steps 2 --warmup 2 --stagger --end-rate 0.003 --no-cpu-baseline --no-exact-fp32 --no-also --no-roofline > $O/prof_steady.log 2>&1; echo "rocprof steady rc=$?"
 f=$(find /tmp/profs_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/steady_kernel_stats.csv)
bash tools/pmc_collect.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
mkdir -p $O/pmc && cp gpurun_out/pmc/*.json $O/pmc/ 2>/dev/null
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
