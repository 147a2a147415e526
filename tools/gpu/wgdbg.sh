#!/bin/bash
# weight-gradient check after a change of wgrad_ps_kernel: its GPU tests, the plan-against-plan comparison, per-shape times against v2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wgdbg
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "wgrad" > gpurun_out/wgdbg/tests.log 2>&1; tail -8 gpurun_out/wgdbg/tests.log | grep -v "^$"
python tools/debug/wgrad_race.py 2>&1 | grep -v "amdgpu.ids\|   by" | tee gpurun_out/wgdbg/race.txt
for lib in diamond_amd/libdiamond_hip.so diamond_amd/ablate/libdiamond_hip_wgv2.so diamond_amd/libdiamond_hip.so diamond_amd/ablate/libdiamond_hip_wgv2.so; do
  echo "== wgrad_bench DIAMOND_LIB=$lib"
  DIAMOND_LIB=$lib timeout 300 python tools/wgrad_bench.py wgrad 30 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/wgdbg/bench.txt
