#!/bin/bash
# The staged few-tile convolution (conv_lat_kernel, DIAMOND_CONV_LATENCY_TILES; diamond_amd/csrc/dmd_conv_lat.hip) on the GPU:
# parity first, then a same-box A/B of the B = 1 frame latency and a kernel census of the graphed frame.  ~7 GPU-minutes.
#   gpurun --timeout 600 -- 'bash tools/gpu/staged_latency.sh'
# Results -> gpurun_out/staged_latency/.  If it wins, the cap becomes a default in dmd_conv_lat_route (and the batch-invariance
# tests pin DIAMOND_CONV_LATENCY_TILES=0 where they compare a small launch bitwise with a large one); if not, the file goes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/staged_latency; mkdir -p $O
DIAMOND_STAGED_TESTS=1 timeout 320 python -m pytest tests/test_gpu_staged.py -q -p no:cacheprovider -k latency > $O/tests.log 2>&1
echo "staged tests rc=$?"; tail -3 $O/tests.log
: > $O/ab.txt
for rep in 1 2; do
  for cap in 0 1 4 16 64; do
    DIAMOND_CONV_LATENCY_TILES=$cap timeout 120 python bench.py --config latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap $cap', 'ms/frame graph', round(d['value'], 3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if 'eager' in k})" | tee -a $O/ab.txt
  done
done
# the 8 x 8 level launch by launch on the 8 x 8-block form of the kernel instead of the fused level (144 us per call at B = 1)
for rep in 1 2; do
  DIAMOND_CONV_LATENCY_TILES=64 DIAMOND_LOWRES_CHAIN=0 timeout 120 python bench.py --config latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap 64, no fused 8x8 level', 'ms/frame graph', round(d['value'], 3))" | tee -a $O/ab.txt
done
# the 32-channel layers of the headline window (configs[1], batch 256) on the same kernel: occupancy instead of the pipeline
for rep in 1 2; do
  for c32 in "0 0" "1000000000 0" "1000000000 1"; do
    set -- $c32
    DIAMOND_CONV_LATENCY_TILES_C32=$1 DIAMOND_CONV_LATENCY_TP=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('configs[1] c32 cap / throughput flavour: $c32', round(d['value'], 1), 'frames/s;', {k: v for k, v in list(r['launch_time_ms'].items())[:6]})" | tee -a $O/ab.txt
  done
done
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_lat && DIAMOND_CONV_LATENCY_TILES=64 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lat -o lat -- \
   python $R/bench.py --config latency > $O/prof.log 2>&1; f=$(find /tmp/prof_lat -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/latency_b1_kernel_stats.csv && head -8 $f)
