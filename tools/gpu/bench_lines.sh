#!/bin/bash
# smoke(), the default bench line and the driver's own invocation (python bench.py --gpus 1 --steps 20 --warmup 5)
set -u
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver.err; echo "bench(driver) rc=$?"; tail -3 $O/bench_driver.err
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
