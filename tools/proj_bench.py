"""A/B of the 1x1 skip projection (two 64-channel sources -> 64) on the exact-fp32 kernel vs the split-fp16 instance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv
dev = "cuda"
for n, h, cins, cout in [(256, 64, [64, 64], 64), (256, 32, [64, 64], 64), (256, 16, [64, 64], 64), (256, 8, [64, 64], 64), (256, 16, [32], 64)]:
    srcs = [(E.Act(torch.randn(n, h, h, c, device=dev)), 0, None) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    wp, w16, b = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w), torch.zeros(cout, device=dev)
    for name, kw in (("f32", {}), ("f16x2", {"w_f16": w16})):
        run = lambda: E.conv2d(srcs, wp, b, cout, taps=1, want_stats=False, **kw)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = 4.0 * n * h * h * (cin + cout)
        print(f"N{n} {h}x{h} cin{cin}->{cout} {name}: {ms*1e3:7.1f} us  {nbytes/ms/1e6:6.0f} GB/s", flush=True)
