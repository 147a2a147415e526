"""Timing of dmd_lowres_chain alone (development aid): the 8x8 level of the default U-Net at batch 256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diamond_amd as D
from diamond_amd import engine as E
from diamond_amd.blocks import FilmTable, RunCtx
from diamond_amd.testing import fill_module_

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
agent = D.Agent(D.default_agent_config())
fill_module_(agent, 7)
agent = agent.cuda().eval()
im = agent.denoiser.inner_model
film = FilmTable(im.unet)
table = film.compute(torch.randn(n, 256, device="cuda"))
ctx = RunCtx(im._cache, film, table, precision="f16x2")
x = E.Act(torch.randn(n, 8, 8, 64, device="cuda"))
with torch.no_grad():
    for _ in range(3):
        im.unet._run_lowres_chain(ctx, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        im.unet._run_lowres_chain(ctx, x)
    e1.record()
    torch.cuda.synchronize()
print(f"lowres chain N={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call", flush=True)

if "--trace" in sys.argv:  # needs a -DDMD_LAB -DLR_TRACE=1 build (DIAMOND_LIB=...)
    import ctypes as C
    from diamond_amd import native as nv
    buf = (C.c_ulonglong * 512)()
    k = nv.lib().dmd_lowres_trace_dump(buf)
    names = {0: "input+stats", 1: "(block start)", 2: "tab1", 3: "stage1", 4: "conv1 mfma", 5: "epi1", 6: "stats H", 7: "proj", 8: "tab2+stage2",
             9: "conv2 mfma", 10: "epi2+stats", 11: "attn x_n", 12: "attn qkv", 13: "attn core", 14: "attn out+stats", 15: "save+store"}
    prev = buf[0] >> 8
    tot = {}
    for i in range(1, k):
        t, tag = buf[i] >> 8, buf[i] & 0xff
        tot.setdefault(tag, []).append(t - prev)
        prev = t
    clk = 1.97e9  # shader clock (s_memtime ticks are shader cycles; ~1.97 GHz measured from the whole-call time)
    for tag in sorted(tot):
        v = tot[tag]
        print(f"{names.get(tag, tag):16s} n={len(v):2d} total {sum(v) / clk * 1e6:7.1f} us  each {sum(v) / len(v) / clk * 1e6:6.2f} us")
    print("whole", (buf[k - 1] >> 8) - (buf[0] >> 8), "ticks")
