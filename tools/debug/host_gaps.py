import os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from diamond_amd import native as nv
import bench
L = nv.lib()
cd = L._cdll
log = []
real = {}
for name in nv.LAUNCHERS:
    try:
        real[name] = getattr(cd, name)
    except AttributeError:
        pass
class Wrap:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); rc = self.fn(*a); t1 = time.perf_counter()
        tag = self.name
        if self.name == "dmd_conv2d":
            p = a[0]._obj
            tag = "conv2d proj" if p.proj_nsrc else f"conv2d c{p.Cout} {p.H}"
        log.append((t0, t1, tag)); return rc
orig_getattr = type(L).__getattr__
def ga(self, name):
    if name in real:
        return Wrap(name, real[name])
    return orig_getattr(self, name)
type(L).__getattr__ = ga
sync_done = []
_any = torch.Tensor.any
def any_logged(self, *a, **k):
    r = _any(self, *a, **k)
    if r.ndim == 0 and r.is_cuda:
        v = bool(r)  # the host sync of `if dead.any():`
        sync_done.append(time.perf_counter())
    return r
torch.Tensor.any = any_logged
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-exact-fp32", "--no-roofline"]
bench.main()
import bisect
starts = [t0 for t0, _, _ in log]
lat, burst = [], []
for t in sync_done[len(sync_done) // 2:]:
    i = bisect.bisect_left(starts, t)
    if i + 45 < len(log):
        lat.append((log[i][0] - t) * 1e3)
        burst.append(((log[i + 40][0] - log[i][0]) * 1e3, [x[2] for x in log[i:i + 6]]))
print("host latency sync-return -> first launch (ms): mean %.3f max %.3f" % (sum(lat) / len(lat), max(lat)))
print("host time for the next 40 launches (ms): mean %.3f" % (sum(b for b, _ in burst) / len(burst)), burst[0][1])
# analyse the last 60 % of the log
log = log[int(0.4 * len(log)):]
long_calls = sorted(((t1 - t0) * 1e3, tag) for t0, t1, tag in log)[-12:]
print("longest launch calls (ms):", [(round(a, 2), b) for a, b in long_calls])
gaps = []
for (a0, a1, at), (b0, b1, bt) in zip(log, log[1:]):
    gaps.append(((b0 - a1) * 1e3, at, bt))
gaps.sort()
print("longest host intervals between launches (ms):")
for g, a, b in gaps[-25:]:
    print(f"  {g:8.2f}  {a}  ->  {b}")
