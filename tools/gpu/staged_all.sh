#!/bin/bash
# First GPU call after a round that ended with staged (unmeasured) kernels: parity of everything staged, then the two same-box
# A/Bs.  ~17 GPU-minutes.     gpurun --timeout 1400 -- 'bash tools/gpu/staged_all.sh'
# Reads: gpurun_out/staged_wgrad/{tests.log,ab.txt}, gpurun_out/staged_latency/{tests.log,ab.txt,latency_b1_kernel_stats.csv}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
(rocm-smi --showproductname --showclocks --showpower 2>/dev/null | head -40) > gpurun_out/staged_box.txt 2>&1 || true
timeout 680 bash tools/gpu/staged_wgrad.sh
timeout 680 bash tools/gpu/staged_latency.sh
