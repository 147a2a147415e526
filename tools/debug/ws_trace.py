"""Timeline of one workgroup of conv_f16ws (s_memtime stamps, WS_TRACE build): who waits for whom in a chunk step.
   DIAMOND_LIB=diamond_amd/ablate/libdiamond_hip_trace.so python tools/debug/ws_trace.py [cin] [res]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = "cuda"
n, h = 256, 64
cins = [64] * (cin // 64)
srcs = []
for c in cins:
    a = E.gn_stats(torch.randn(n, h, h, c, device=dev))
    spec = E.NormSpec(mul=torch.randn(n, c, device=dev) * 0.1, add=torch.randn(n, c, device=dev) * 0.1, mul_stride=c, add_stride=c, plus_one=True)
    srcs.append((a, 1, spec))
w = torch.randn(64, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
wp, w16 = nv.pack_conv_weight(w), nv.pack_conv_weight_f16x2(w)
b = torch.zeros(64, device=dev)
r = E.Act(torch.randn(n, h, h, 64, device=dev)) if res else None
run = lambda: E.conv2d(srcs, wp, b, 64, residual=r, w_f16=w16)
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
print(f"launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
L.dmd_ws_trace_dump(buf, cnt)
ev = {}
for role in range(3):
    for i in range(cnt[role]):
        v = buf[role * NMAX + i]
        ev.setdefault(role, []).append((v >> 16, (v >> 12) & 0xf, v & 0xfff))
t0 = min(e[0][0] for e in ev.values())
names = {0: "own:start", 1: "own:mfma_done", 2: "own:after_barrier", 8: "oth:start", 9: "oth:epi_done", 10: "oth:after_barrier"}
pn = {0: "P:start", 1: "P:loadW_issued", 2: "P:storeS_done", 3: "P:issueS_done", 4: "P:storeW_done", 5: "P:after_barrier"}
print("counts", list(cnt))
# per-step summary for steps 8..40
import collections
rows = collections.defaultdict(dict)
for role, lst in ev.items():
    for t, tag, step in lst:
        rows[step][(role, tag)] = t - t0
steps = sorted(rows)
def d(step, a, b):
    r = rows[step]
    return (r[b] - r[a]) if (a in r and b in r) else None
print("step | consumer: mfma_phase barrier_wait | epi-group: epi barrier_wait | producer: loadW storeS issueS storeW barrier_wait | step_total(cycles of memtime @100MHz?)")
tot = collections.Counter(); cntr = collections.Counter()
for s in steps[8:72]:
    own = 0 if (0, 0) in rows[s] else 1
    oth = 1 - own
    vals = {
        "mfma": d(s, (own, 0), (own, 1)), "c_wait": d(s, (own, 1), (own, 2)),
        "epi": d(s, (oth, 8), (oth, 9)), "e_wait": d(s, (oth, 9), (oth, 10)),
        "loadW": d(s, (2, 0), (2, 1)) if (2, 0) in rows[s] else None, "storeS": d(s, (2, 1), (2, 2)), "issueS": d(s, (2, 2), (2, 3)),
        "storeW": d(s, (2, 3), (2, 4)), "p_wait": d(s, (2, 4), (2, 5)),
    }
    nxt = rows.get(s + 1, {})
    tot_step = None
    if (2, 5) in rows[s] and (2, 5) in rows.get(s - 1, {}):
        tot_step = rows[s][(2, 5)] - rows[s - 1][(2, 5)]
    vals["step"] = tot_step
    for k, v in vals.items():
        if v is not None:
            tot[k] += v; cntr[k] += 1
    if s < 30:
        print(s, {k: v for k, v in vals.items()})
print("MEAN over steps:", {k: round(tot[k] / cntr[k], 1) for k in tot})
span = max(e[-1][0] for e in ev.values()) - t0
print("span ticks", span, "-> tick/us =", span / (e0.elapsed_time(e1) * 1e3))
