#!/bin/bash
# One GPU-box session: the whole -m gpu suite, idle-gap traces of three regimes (tools/gpu/gaps.sh) and the env-loop A/B (tools/ab_env_loop.py)
set -u
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 > $O/tests.log 2>&1
echo "pytest rc=$?" | tee -a $O/tests.log; tail -6 $O/tests.log
bash tools/gpu/gaps.sh $TAG/gaps
timeout 400 python tools/ab_env_loop.py > $O/ab_env_loop.json 2> $O/ab_env_loop.err; echo "ab rc=$?"; grep "^\[1\]" $O/ab_env_loop.err
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > $O/box.txt
