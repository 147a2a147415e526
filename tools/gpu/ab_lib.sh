#!/bin/bash
# A library build against the shipping one on one box (DIAMOND_LIB): output checksums of fixed-seed convolutions (bit-identity),
# per-shape timings and the bench window, alternating.   bash tools/gpu/ab_lib.sh <tag> diamond_amd/ablate/libdiamond_hip_X.so
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-ablib}; mkdir -p $O
LIB=$2
export TMPDIR=/tmp
for lib in diamond_amd/libdiamond_hip.so $LIB; do
  echo "== checksum $lib" | tee -a $O/ab_lib.txt
  DIAMOND_LIB=$lib python - <<'PY' 2>&1 | tee -a $O/ab_lib.txt
import sys, torch, hashlib
sys.path.insert(0, ".")
from diamond_amd import engine as E, native as nv
g = torch.Generator(device="cuda").manual_seed(5)
for n, h, cins, res in ((64, 64, [64], True), (32, 32, [64, 64], False), (16, 64, [32], True)):
    cout = 32 if cins == [32] else 64
    srcs = []
    for c in cins:
        x = torch.randn(n, h, h, c, device="cuda", generator=g) * 1.5 + 0.3
        spec = E.NormSpec(mul=torch.randn(n, c, device="cuda", generator=g) * 0.3, add=torch.randn(n, c, device="cuda", generator=g) * 0.3, mul_stride=c, add_stride=c, plus_one=True)
        srcs.append((E.gn_stats(x), 1, spec))
    w = torch.randn(cout, sum(cins), 3, 3, device="cuda", generator=g) / (sum(cins) * 9) ** 0.5
    r = E.Act(torch.randn(n, h, h, cout, device="cuda", generator=g)) if res else None
    out = E.conv2d(srcs, nv.pack_conv_weight(w), torch.zeros(cout, device="cuda"), cout, residual=r, w_f16=nv.pack_conv_weight_f16x2(w))
    torch.cuda.synchronize()
    print(n, h, cins, hashlib.sha1(out.t.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(out.stats.cpu().numpy().tobytes()).hexdigest()[:16])
PY
done
for lib in diamond_amd/libdiamond_hip.so $LIB diamond_amd/libdiamond_hip.so $LIB; do
  echo "== conv_bench $lib" | tee -a $O/ab_lib.txt
  DIAMOND_LIB=$lib timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -6 | tee -a $O/ab_lib.txt
done
STEPS=3 bash tools/gpu/ab_bench.sh DIAMOND_LIB diamond_amd/libdiamond_hip.so $LIB 2>&1 | tee -a $O/ab_lib.txt
