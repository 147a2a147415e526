#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -s -p no:cacheprovider -k "training_step" > $O/tests.log 2>&1; tail -40 $O/tests.log | cut -c1-1500
