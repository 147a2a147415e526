#!/bin/bash
# Same-box A/B of the long-sequence attention kernel between two builds of the library: parity tests first, then the kernel at
# configs[4]'s shapes, then bench.py --config 4, alternating.
#   gpurun --timeout 900 -- 'bash tools/gpu/ab_attention.sh diamond_amd/ablate/libdiamond_hip_pre_max3.so'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
OLD=$1; NEW=diamond_amd/libdiamond_hip.so
O=gpurun_out/ab_attention; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x -k "attention or attn or 256" 2>&1 | tail -3 | tee $O/tests.txt
for lib in $OLD $NEW $OLD $NEW; do
  echo "== $lib" | tee -a $O/kernel.txt
  DIAMOND_LIB=$lib timeout 120 python tools/attention_bench.py 4096 1024 2>&1 | tail -2 | tee -a $O/kernel.txt
done
for lib in $OLD $NEW $OLD $NEW; do
  echo "== $lib" | tee -a $O/cfg4.txt
  DIAMOND_LIB=$lib timeout 300 python bench.py --config 4 --no-cpu-baseline --no-exact-fp32 --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), d['unit'], d['ms_per_step'], 'ms per step')" | tee -a $O/cfg4.txt
done
