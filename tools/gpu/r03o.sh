#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03o; mkdir -p $O
timeout 300 python bench.py --config train 2>$O/train.err | tee $O/train.json; tail -3 $O/train.err | cut -c1-300
timeout 300 python bench.py --config train --batch 256 --steps 5 2>$O/train256.err | tee $O/train256.json; tail -3 $O/train256.err | cut -c1-300
for t in 0 6 24 80; do echo "== latency MIN_TILES=$t"; DIAMOND_WS_MIN_TILES=$t timeout 300 python bench.py --config latency --steps 100 2>$O/lat_$t.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v,3) for k,v in d.items() if isinstance(v,float)})"; done
