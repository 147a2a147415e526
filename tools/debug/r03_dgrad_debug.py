"""Round-3 diagnostic: conv_f16ws on gradient-like inputs (wide dynamic range, no prologue) vs fp64, per geometry."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from diamond_amd import engine as E, native as nv

dev = "cuda"
g = torch.Generator().manual_seed(3)
for (n, h, cin, cout, taps, mode) in [(256, 8, 64, 64, 9, "grad"), (256, 8, 64, 64, 9, "unit"), (64, 16, 64, 64, 9, "grad"), (64, 16, 64, 32, 9, "grad"),
                                      (256, 8, 64, 32, 9, "grad"), (256, 16, 64, 32, 9, "grad"), (16, 64, 32, 32, 9, "grad"), (256, 8, 64, 64, 9, "sparse")]:
    x = torch.randn(n, h, h, cin, generator=g)
    if mode == "grad":
        x = x * torch.pow(10.0, -6 * torch.rand(n, h, h, cin, generator=g))
    if mode == "sparse":
        x = x * (torch.rand(n, h, h, cin, generator=g) < 0.05) * torch.pow(10.0, -4 * torch.rand(n, h, h, cin, generator=g))
    x = x / x.abs().max()
    w = torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) / (cin * taps) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1 if taps == 9 else 0).permute(0, 2, 3, 1)
    wd = w.to(dev)
    out = E.conv2d([(E.Act(x.to(dev).contiguous()), nv.PROLOGUE_NONE, None)], nv.pack_conv_weight(wd), None, cout, taps=taps, want_stats=False,
                   w_f16=nv.pack_conv_weight_f16x2(wd)).t.cpu().double()
    err = (out - ref).abs()
    print(f"N{n} {h}x{h} {cin}->{cout} taps{taps} {mode}: max err / max ref {float(err.max() / ref.abs().max()):.2e}   rms err / rms ref {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}", flush=True)
