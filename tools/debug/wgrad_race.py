"""Where a wrong weight gradient differs from the right one (development aid): cap 256 = reference, cap 512 = the plan under test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import ac_native as A, engine as E, native as nv


def setcap(c):
    os.environ["DIAMOND_WGRAD_MAX_WG"] = c
    nv.lib().dmd_reload_env()

DEV = "cuda"
n, h, w, cin, cout, taps = int(os.environ.get("RACE_N", "96")), 64, 64, 32, 32, 9
for prologue in (0, 1, 2):
    g = torch.Generator().manual_seed(n + cin)
    x = (torch.randn(n, h, w, cin, generator=g) * 1.3 + 0.2).to(DEV)
    dy = torch.randn(n, h, w, cout, generator=g).to(DEV)
    xa = E.gn_stats(x) if prologue else E.Act(x)
    spec = E.NormSpec(mul=(torch.randn(cin, generator=g) * 0.2 + 1).to(DEV), add=(torch.randn(cin, generator=g) * 0.2).to(DEV)) if prologue else None
    setcap("256")
    ref, refb = A._wgrad(xa, prologue, spec, dy, taps, cin, split=True)
    torch.cuda.synchronize()
    for cap in ("256", "384") + ("512",) * int(os.environ.get("RACE_REPS", "3")):
        setcap(cap)
        dw, db = A._wgrad(xa, prologue, spec, dy, taps, cin, split=True)
        torch.cuda.synchronize()
        d = (dw - ref)
        bad = d != 0
        print(f"prologue {prologue} cap {cap}: {int(bad.sum())} of {bad.numel()} entries differ, max |d| {float(d.abs().max()):.3e} (|ref| max {float(ref.abs().max()):.3e}); db differs {int((db != refb).sum())}")
        if bad.any():
            print("   by tap:", bad.sum(dim=(0, 1)).flatten().tolist())
            print("   by cout:", bad.sum(dim=(1, 2, 3)).tolist())
            print("   by cin:", bad.sum(dim=(0, 2, 3)).tolist())
