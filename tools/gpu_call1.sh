#!/bin/bash
# round 5, first GPU call: the new host-side paths (pipelined env loop, reference drivers, 200k-pixel bars, self-launch), the env
# loop A/B under episodes that end, and the default bench line.
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -s -p no:cacheprovider --durations=10 tests/test_gpu_reference_drivers.py tests/test_gpu_dist.py tests/test_gpu_env.py \
   "tests/test_gpu_models.py" -k "pipelined or speculative or full_window or 200k or reference or dist or ring or window or batch_shard or self_launch or world1" > $O/tests_new.log 2>&1
echo "pytest(new) rc=$?" | tee -a $O/tests_new.log; tail -5 $O/tests_new.log
timeout 500 python tools/ab_env_loop.py > $O/ab_env_loop.json 2> $O/ab_env_loop.err; echo "ab rc=$?"; tail -40 $O/ab_env_loop.err
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench_line.json; tail -5 $O/bench.err
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
