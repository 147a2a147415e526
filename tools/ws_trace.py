"""Timeline of one workgroup of conv_f16ws_kernel (s_memtime stamps of a DMD_LAB -DWS_TRACE build): who waits for whom
in a chunk step.
   bash tools/build_ws_ablations.sh trace   (WS_EXTRA=-DWS_TRACE)
   DIAMOND_LIB=diamond_amd/ablate/libdiamond_hip_wstrace.so python tools/ws_trace.py [cin] [res] [cout] [h]
   res = 2: the fused skip projection (WsGeomProj) instead of a residual; cout = 32: the 512-pixel-tile instance
   (WsGeom<false, 1, 9>: reward / end model, actor-critic); h: image size (64)"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = "cuda"
h = int(sys.argv[4]) if len(sys.argv) > 4 else 64
n = int(sys.argv[5]) if len(sys.argv) > 5 else 256  # n < 38: workgroup 0 is traced, and every stamp is listed
srcs = []
for c in ([64] * (cin // 64) if cin >= 64 else [cin]):
    a = E.gn_stats(torch.randn(n, h, h, c, device=dev))
    spec = E.NormSpec(mul=torch.randn(n, c, device=dev) * 0.1, add=torch.randn(n, c, device=dev) * 0.1, mul_stride=c, add_stride=c, plus_one=True)
    srcs.append((a, 1, spec))
w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
wp, w16 = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w)
b = torch.zeros(cout, device=dev)
r = E.Act(torch.randn(n, h, h, cout, device=dev)) if res == 1 else None
proj = None
if res == 2:
    wpj = torch.randn(64, 128, 1, 1, device=dev) / 128 ** 0.5
    proj = ([E.Act(torch.randn(n, h, h, 64, device=dev)), E.Act(torch.randn(n, h, h, 64, device=dev))], nv.pack_conv_weight_f16x2(wpj), b)
run = lambda: E.conv2d(srcs, wp, b, cout, residual=r, w_f16=w16, proj=proj)
L = nv.lib()
L.dmd_ws_trace_dump.argtypes = [C.c_void_p, C.c_void_p]
NMAX = 4096
buf = (C.c_ulonglong * (3 * NMAX))()
cnt = (C.c_int * 3)()
for _ in range(3):
    run()
L.dmd_ws_trace_dump(buf, cnt)  # reset
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
L.dmd_ws_trace_dump(buf, cnt)
rows = collections.defaultdict(dict)  # step -> {(role, tag): tick}
tmin, tmax = None, 0
for role in range(3):
    for i in range(cnt[role]):
        v = buf[role * NMAX + i]
        t, tag, step = v >> 16, (v >> 12) & 0xf, v & 0xfff
        rows[step][(role, tag)] = t
        tmin = t if tmin is None else min(tmin, t)
        tmax = max(tmax, t)
print(f"launch {us:.1f} us, counts {list(cnt)}, span {tmax - tmin} ticks = {(tmax - tmin) / us:.0f} ticks/us")
if n < 38:  # the raw timeline of a few-tile launch (tags 13: kernel entry / in front of / behind the tail write-out, 14 / 15: in front of / behind B(-1), 12: in front of B0)
    ev = sorted((rows[s][k], k[0], k[1], s) for s in rows for k in rows[s])
    for t, role, tag, step in ev:
        print(f"  +{t - tmin:7d} ticks  role {role} tag {tag:2d} step {step}")
# producer: 0 start, 7 register set landed, 1 staged, 2 issued, 3 past barrier.  active consumer: 4 start, 5 body done, 6 past barrier.
# write-out consumer: 8 start, 9 DMA issued, 10 epilogue + landing done, 11 past barrier
tot, num = collections.Counter(), collections.Counter()
def d(r, a, b):
    return r[b] - r[a] if a in r and b in r else None
for s in sorted(rows)[4:-4]:
    r = rows[s]
    act = 0 if (0, 4) in r else 1
    oth = 1 - act
    vals = {"P.await": d(r, (2, 0), (2, 7)), "P.math": d(r, (2, 7), (2, 1)), "P.issue": d(r, (2, 1), (2, 2)), "P.wait": d(r, (2, 2), (2, 3)),
            "C.mfma": d(r, (act, 4), (act, 5)), "C.wait": d(r, (act, 5), (act, 6)),
            "W.proj0": d(r, (oth, 8), (oth, 12)), "W.dma": d(r, (oth, 12) if (oth, 12) in r else (oth, 8), (oth, 9)), "W.epi": d(r, (oth, 9), (oth, 10)), "W.wait": d(r, (oth, 10), (oth, 11))}
    prev = rows.get(s - 1, {})
    if (2, 3) in r and (2, 3) in prev:
        vals["step"] = r[(2, 3)] - prev[(2, 3)]
    for k, v in vals.items():
        if v is not None:
            tot[k] += v
            num[k] += 1
    if 8 <= s < 24:
        print(s, {k: v for k, v in vals.items() if v is not None})
print("MEAN ticks:", {k: round(tot[k] / num[k]) for k in tot})
