#!/bin/bash
# WsGeomW64 (DIAMOND_WS_W64=1: 64-cout x 64-pixel consumer wave tile) against the shipping geometry, one box: parity at the
# production launch configurations, per-shape timings (alternating), and the bench window (alternating).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-w64}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tpw.py -m gpu -q -p no:cacheprovider -k "w64" > $O/tests_w64.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests_w64.log
for v in 0 1 0 1; do
  echo "== conv_bench DIAMOND_WS_W64=$v" | tee -a $O/conv_bench.txt
  DIAMOND_WS_W64=$v timeout 300 python tools/conv_bench.py 2>&1 | tee -a $O/conv_bench.txt
done
STEPS=3 bash tools/gpu/ab_bench.sh DIAMOND_WS_W64 0 1 2>&1 | tee $O/ab_window.txt
