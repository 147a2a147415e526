#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_graph.py "tests/test_gpu_models.py::test_denoiser_training_step_vs_reference_golden" -x -q -s 2>&1 | tail -25 > $O/tests.log; cat $O/tests.log | tail -14
timeout 300 python bench.py --config train 2>$O/train.err | tee $O/train.json; tail -3 $O/train.err
timeout 300 python bench.py --config train --batch 256 --steps 5 2>$O/train256.err | tee $O/train256.json; tail -3 $O/train256.err
timeout 300 python bench.py --config latency 2>$O/lat.err | tee $O/lat.json; tail -3 $O/lat.err
