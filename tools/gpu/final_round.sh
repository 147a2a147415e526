#!/bin/bash
# The whole -m gpu suite, smoke(), the driver's bench invocation, rocprofv3 kernel stats of a bench run and the PMC traffic passes.
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- \
   python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
 f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -5 $f | cut -c1-160)
bash tools/pmc_collect.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
mkdir -p $O/pmc && cp gpurun_out/pmc/*.json $O/pmc/ 2>/dev/null
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
