"""One conv_f16ws launch shape in a loop for a given time (development aid for tools/gpu/clock_table.sh: something for the
clock / power sensors to look at).    python tools/conv_loop.py <seconds> [zero] [cin] [cout] [res] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
zero = len(sys.argv) > 2 and sys.argv[2] == "1"
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cout = int(sys.argv[4]) if len(sys.argv) > 4 else 64
res = len(sys.argv) > 5 and sys.argv[5] == "1"
dev, n, h = "cuda", (int(sys.argv[6]) if len(sys.argv) > 6 else 256), 64
mk = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, device=dev))
srcs = []
for c in ([64] * (cin // 64) if cin >= 64 else [cin]):
    a = E.gn_stats(mk(n, h, h, c))
    spec = E.NormSpec(mul=mk(n, c) * 0.1, add=mk(n, c) * 0.1, mul_stride=c, add_stride=c, plus_one=True)
    srcs.append((a, 1, spec))
w = mk(cout, cin, 3, 3) / (cin * 9) ** 0.5
wp, w16 = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w)
b = torch.zeros(cout, device=dev)
r = E.Act(mk(n, h, h, cout)) if res else None
run = lambda: E.conv2d(srcs, wp, b, cout, residual=r, w_f16=w16)
for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
per = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    per.append(e0.elapsed_time(e1) / 50 * 1e3)
per.sort()
flops = 2.0 * n * h * h * cout * cin * 9
print(f"conv {cin}->{cout} 64x64 B{n} res{int(res)} zero{int(zero)}: median {per[len(per) // 2]:.1f} us per launch (min {per[0]:.1f}, max {per[-1]:.1f}, "
      f"{len(per) * 50} launches), {flops / per[len(per) // 2] / 1e6:.0f} TFLOP/s algorithmic")
