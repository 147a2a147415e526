"""Micro-benchmark of the backward's own kernels (development aid, GPU): dmd_conv2d_wgrad and dmd_gn_silu_bwd at the shapes of the
denoiser training step (batch 32: 64x64 ... 8x8 levels, 64 channels) and of the actor-critic backward (3,840 frames, 32 / 64
channels), microseconds per call (HIP events around REPS calls on the launch stream) and a checksum of the result so that two
libraries can be compared (`DIAMOND_LIB=... python tools/wgrad_bench.py`).
usage: python tools/wgrad_bench.py [wgrad|gn|all] [reps] [only: a substring of the shape's name]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from diamond_amd import ac_native as A, engine as E, native as nv

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
only = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("WGRAD_ONLY", "")  # (the env form: for wrappers that cannot quote)
nv.PRECISION_F16X2 |= int(os.environ.get("WGRAD_LAB", "0")) << 8  # (a library built with EXTRA_HIPCC_FLAGS=-DDMD_LAB only: 1 no contraction, 2 no staging, 4 no prefetch loads)
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(1)


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps, out


WG = [  # name, N, H, Cin, Cout, taps, prologue
    ("f2 64^2 64->64", 32, 64, 64, 64, 9, 1), ("f2 32^2 64->64", 32, 32, 64, 64, 9, 1), ("f2 16^2 64->64", 32, 16, 64, 64, 9, 1),
    ("f2 64^2 64->64 raw", 32, 64, 64, 64, 9, 0), ("f2 64^2 1x1", 32, 64, 64, 64, 1, 0),
    ("ac 64^2 32->32", 3840, 64, 32, 32, 9, 1), ("ac 32^2 32->32", 3840, 32, 32, 32, 9, 1), ("ac 16^2 32->64", 3840, 16, 32, 64, 9, 1),
    ("ac 8^2 64->64", 3840, 8, 64, 64, 9, 1),
]
if what in ("wgrad", "all"):
    for name, n, h, cin, cout, taps, pro in WG:
        if only not in name:
            continue
        x = torch.randn(n, h, h, cin, device=dev, generator=g) * 1.3 + 0.2
        dy = torch.randn(n, h, h, cout, device=dev, generator=g)
        xa = E.gn_stats(x)
        spec = E.NormSpec(mul=torch.randn(cin, device=dev, generator=g) * 0.2 + 1, add=torch.randn(cin, device=dev, generator=g) * 0.2) if pro else None
        us, (dw, db) = timed(lambda: A._wgrad(xa, pro, spec, dy, taps, cin, split=True))
        flop = 2.0 * taps * cin * cout * n * h * h
        print(f"wgrad {name:22s} {us:8.1f} us  {flop/us/1e6:7.1f} TFLOP/s algorithmic   sum {float(dw.double().sum()):+.9e} abs {float(dw.double().abs().sum()):.9e}", flush=True)
        del x, dy, xa
GN = [("f2 64^2 c64", 32, 64, 64), ("f2 32^2 c64", 32, 32, 64), ("f2 64^2 c128", 32, 64, 128), ("ac 64^2 c32", 3840, 64, 32), ("ac 32^2 c32", 3840, 32, 32),
      ("ac 16^2 c64", 3840, 16, 64)]
if what in ("gn", "all"):
    for name, n, h, c in GN:
        if only not in name:
            continue
        x = torch.randn(n, h, h, c, device=dev, generator=g) * 1.7 + 0.4
        da = torch.randn(n, h, h, c, device=dev, generator=g)
        dskip = torch.randn(n, h, h, c, device=dev, generator=g)
        xa = E.gn_stats(x)
        spec = E.NormSpec(mul=torch.randn(c, device=dev, generator=g) * 0.2 + 1, add=torch.randn(c, device=dev, generator=g) * 0.2)
        us, (dx, dmul, dadd) = timed(lambda: A._gn_silu_bwd(xa, spec, da, dskip))
        mb = 4.0 * x.numel() * 4 / 1e6
        print(f"gn_bwd {name:22s} {us:8.1f} us  {mb/us*1e3:7.1f} GB/s algorithmic ({mb:.0f} MB)   sum {float(dx.double().sum()):+.9e} dmul {float(dmul.double().sum()):+.9e}", flush=True)
        del x, da, dskip, xa
