#!/bin/bash
# One GPU-box session: parity tests, the bench lines, rocprofv3 kernel stats and the PMC traffic passes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r03'
# Everything lands in gpurun_out/<tag>/ (copy what should be judged into profiles/ afterwards).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 ${PYTEST_ARGS:-} > $O/tests.log 2>&1
  echo "pytest rc=$?" >> $O/tests.log
  tail -3 $O/tests.log
fi
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench_line.json
if [ "${SKIP_PROFILE:-0}" != "1" ]; then
  (cd /tmp && rm -rf /tmp/prof_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- \
     python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
   f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -12 $f)
  bash tools/pmc_collect.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
  mkdir -p $O/pmc && cp gpurun_out/pmc/*.json $O/pmc/ 2>/dev/null
  # configs[4] (256x256, attention): kernel stats of one window + the same two PMC passes
  (cd /tmp && rm -rf /tmp/prof4_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4_$TAG -o bench -- \
     python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_cfg4.log 2>&1
   f=$(find /tmp/prof4_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv && head -8 $f)
  CFG=4 bash tools/pmc_collect.sh > $O/pmc_cfg4.log 2>&1; echo "pmc cfg4 rc=$?"
  mkdir -p $O/pmc_cfg4 && cp gpurun_out/pmc_cfg4/*.json $O/pmc_cfg4/ 2>/dev/null
fi
if [ "${SKIP_CONFIGS:-0}" != "1" ]; then
  timeout 500 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; echo "cfg4 rc=$?"; tail -c 300 $O/bench_config4.json
  timeout 400 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "cfg3 rc=$?"; tail -c 300 $O/bench_config3.json
  timeout 300 python bench.py --config latency > $O/bench_latency.json 2> $O/bench_latency.err; echo "latency rc=$?"; tail -c 400 $O/bench_latency.json
  timeout 300 python bench.py --config train > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"; tail -c 400 $O/bench_train.json
  timeout 300 python bench.py --config train --batch 256 --steps 5 >> $O/bench_train.json 2>> $O/bench_train.err
fi
# which box: GPU name / clocks as the driver reports them (boxes differ by a few per cent at identical code)
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
