#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_models.py -m gpu -q -s -p no:cacheprovider -k "rccl or bench_distributed or heun5 or lowres_chain" > $O/tests.log 2>&1; tail -3 $O/tests.log; grep -E "^FAILED|^ERROR|heun5 step|ddp_grad" $O/tests.log | head -20
