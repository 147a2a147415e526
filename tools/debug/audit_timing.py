"""Where does WeightAudit.run() spend host time?  (round 6: an 18 ms hole per window sat between its `!=` and `.any` kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_agent
from diamond_amd import engine as E
from diamond_amd.testing import synthetic_actions, synthetic_frames

dev = torch.device("cuda:0")
agent = build_agent(dev, 64, 0)
g = torch.Generator().manual_seed(0)
obs = synthetic_frames(g, 256, 12, 64, 64).to(dev)
act = synthetic_actions(g, 4, 256, 4).to(dev)
x = torch.randn(256, 3, 64, 64, device=dev)
for _ in range(3):
    agent.denoiser.denoise(x, 1.0, obs, act)
torch.cuda.synchronize()
audits = [a for c in list(E._WEIGHT_CACHES) for a in c.audits()]
print("audits:", [(a.what, len(a._refs)) for a in audits])
for rep in range(4):
    for a in audits:
        if a._table is None or not a._refs:
            continue
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        a.check()
        n = len(a._refs)
        believed = [r() is not None and a._stamps[i] is not None and a._stamps[i] == E._stamp(r()) for i, r in enumerate(a._refs)]
        t.append(time.perf_counter())
        a._launch(0, n, a._live)
        t.append(time.perf_counter())
        ne = a._live[:n] != a._rec[:n]
        t.append(time.perf_counter())
        bad = ne.any(dim=1)
        t.append(time.perf_counter())
        host = torch.empty(n, dtype=torch.bool).pin_memory()
        t.append(time.perf_counter())
        host.copy_(bad, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        t.append(time.perf_counter())
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        names = ["stamps", "launch", "ne", "any", "pin_memory", "copy+event", "sync"]
        print(rep, a.what, n, {k: round(1e3 * (b - a_), 3) for k, a_, b in zip(names, t, t[1:])})
