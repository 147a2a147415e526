#!/bin/bash
# Last call of a round: the -m gpu suite and smoke() on the final tree, the default bench line, and the configs[4] line (with its
# PMC traffic and the reference CPU baseline at 256x256).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench_line.json
timeout 500 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; echo "cfg4 rc=$?"; tail -c 400 $O/bench_config4.json
