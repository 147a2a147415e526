"""A chain of conv_f16ws launches in which every launch reads what the previous one wrote (development aid): the producer ->
consumer pattern of the U-Net's level 0, where conv_loop.py (one launch re-reading the same input) says nothing about what
is still in the 256 MB Infinity Cache when the consumer starts.
    python tools/conv_chain.py <seconds> [n] [cin] [cout] [nbuf]
n: images (256); nbuf: activation buffers cycled through (2 = ping-pong).  Prints us per launch and per image."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cout = int(sys.argv[4]) if len(sys.argv) > 4 else 64
nbuf = int(sys.argv[5]) if len(sys.argv) > 5 else 2
assert cin == cout, "a chain feeds its output back in"
dev, h = "cuda", 64
c = cin
w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
wp, w16 = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w)
b = torch.zeros(cout, device=dev)
spec = E.NormSpec(mul=torch.randn(n, c, device=dev) * 0.1, add=torch.randn(n, c, device=dev) * 0.1, mul_stride=c, add_stride=c, plus_one=True)
x = E.gn_stats(torch.randn(n, h, h, c, device=dev))
keep = [x]


def step(a):
    y = E.conv2d([(a, 1, spec)], wp, b, cout, w_f16=w16)  # GroupNorm + FiLM + SiLU prologue on the producer's statistics
    keep.append(y)
    if len(keep) > nbuf:
        keep.pop(0)  # the caching allocator hands the oldest buffer out again
    return y


for _ in range(6):
    x = step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
per = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        x = step(x)
    e1.record()
    torch.cuda.synchronize()
    per.append(e0.elapsed_time(e1) / 40 * 1e3)
per.sort()
m = per[len(per) // 2]
assert torch.isfinite(x.t).all()
print(f"chain {cin}->{cout} 64x64 N={n} nbuf={nbuf} rev={os.environ.get('DIAMOND_WS_REVERSE', '0')}: median {m:.1f} us per launch = {m / n * 1e3:.1f} ns per image "
      f"(min {per[0]:.1f}), {2.0 * n * h * h * cout * cin * 9 / m / 1e6:.0f} TFLOP/s algorithmic, tensor {n * h * h * c * 4 / 2**20:.0f} MiB")
