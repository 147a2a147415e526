#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_env.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python tools/latency_bench.py 200 > $O/latency.json 2> $O/latency.err; cat $O/latency.json; tail -5 $O/latency.err
