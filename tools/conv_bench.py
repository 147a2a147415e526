"""Per-shape timing of dmd_conv2d (development aid): python tools/conv_bench.py [f16x2|f32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
ZERO = os.environ.get("CONV_BENCH_ZERO", "0") == "1"  # zero-filled activations / weights: the DVFS give-back (same instructions, less toggling)
dev = "cuda"
SHAPES = [  # N, H, [Cins], prologue, residual, upsample
    (256, 64, [64], 1, True, False),
    (256, 64, [64], 1, False, False),
    (256, 64, [64, 64], 1, False, False),
    (256, 64, [64], 0, False, True),
    (256, 32, [64], 1, True, False),
    (256, 32, [64, 64], 1, False, False),
    (256, 16, [64], 1, True, False),
    (256, 16, [64, 64], 1, False, False),
    (256, 8, [64], 1, True, False),
    (256, 8, [64, 64], 1, False, False),
]
COUT = int(os.environ.get("CONV_BENCH_COUT", "64"))
if COUT == 32:
    SHAPES = [(256, 64, [32], 1, True, False), (256, 64, [16], 0, False, False), (256, 32, [32], 1, True, False),
              (256, 16, [32], 1, True, False)]
NB = int(os.environ.get("CONV_BENCH_N", "0"))  # another batch than 256 (the B = 1 latency regime)
for n, h, cins, prologue, res, up in SHAPES:
    n = NB or n
    hs = h // 2 if up else h
    srcs = []
    for c in cins:
        xin = torch.zeros(n, hs, hs, c, device=dev) if ZERO else torch.randn(n, hs, hs, c, device=dev)
        a = E.gn_stats(xin) if prologue else E.Act(xin)
        spec = E.NormSpec(mul=torch.randn(n, c, device=dev) * 0.1, add=torch.randn(n, c, device=dev) * 0.1, mul_stride=c,
                          add_stride=c, plus_one=True) if prologue else None
        srcs.append((a, prologue, spec))
    cin = sum(cins)
    w = torch.randn(COUT, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    if ZERO:
        w.zero_()
    wp = nv.pack_conv_weight(w)
    w16 = nv.pack_conv_weight_f16x2(w) if prec == "f16x2" else None
    b = torch.zeros(COUT, device=dev)
    r = E.Act(torch.randn(n, h, h, COUT, device=dev)) if res else None
    run = lambda: E.conv2d(srcs, wp, b, COUT, upsample=up, residual=r, w_f16=w16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if not NB else 200
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n * h * h * COUT * cin * 9
    nbytes = 4.0 * (n * hs * hs * cin + n * h * h * COUT * (2 if res else 1))
    print(f"N{n} {h}x{h} cin{cin} pro{prologue} res{int(res)} up{int(up)}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s  {nbytes/ms/1e6:7.0f} GB/s", flush=True)
