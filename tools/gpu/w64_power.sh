#!/bin/bash
# Package power / clock of the 64 -> 64, 64x64, batch 256 convolution under the shipping consumer wave tile and under WsGeomW64
# (DIAMOND_WS_W64=1), sensors sampled at 20 Hz while the launch loops for 8 s (tools/smi_sampler.py), alternating.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-w64power}; mkdir -p $O
for v in 0 1 0 1; do
  python tools/smi_sampler.py $O/w64_$v.csv 20 > $O/smi.json 2>/dev/null &
  sp=$!
  sleep 0.5
  DIAMOND_WS_W64=$v timeout 120 python tools/conv_loop.py 8 0 64 64 0 > $O/loop.log 2>&1
  kill $sp; wait $sp 2>/dev/null
  echo "== DIAMOND_WS_W64=$v: $(tail -1 $O/loop.log)  $(cat $O/smi.json)" | tee -a $O/w64_power.txt
done
