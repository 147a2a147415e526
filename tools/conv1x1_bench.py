"""Timing of the streaming 1x1 skip projection (development aid): python tools/conv1x1_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

dev = "cuda"
for n, h, cins in [(256, 64, [64, 64]), (256, 32, [64, 64]), (256, 16, [64, 64])]:
    srcs = [(E.Act(torch.randn(n, h, h, c, device=dev)), nv.PROLOGUE_NONE, None) for c in cins]
    cin = sum(cins)
    w = torch.randn(64, cin, 1, 1, device=dev) / cin ** 0.5
    wp, w16, b = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w), torch.zeros(64, device=dev)
    run = lambda: E.conv2d(srcs, wp, b, 64, taps=1, want_stats=False, w_f16=w16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    nbytes = 4.0 * n * h * h * (cin + 64)
    print(f"N{n} {h}x{h} cin{cin}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
