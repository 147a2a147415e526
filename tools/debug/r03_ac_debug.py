"""Round-3 diagnostic: actor-critic encoder at B=256 -- pooled activations / argmax / gradients saved per library, then compared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

if sys.argv[1] == "run":
    import diamond_amd as D
    from diamond_amd import ac_native
    from diamond_amd.testing import fill_module_, synthetic_frames
    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, 5)
    ac = agent.actor_critic
    g = torch.Generator().manual_seed(258)
    b = 256
    obs = synthetic_frames(g, b, 3, 64, 64)
    wfeat = torch.randn(b, 1024, generator=g) / (b * 15)
    rec = []
    orig = ac_native._maxpool
    def mp(y):
        out, arg = orig(y)
        rec.append((y.detach().cpu(), arg.detach().cpu()))
        return out, arg
    ac_native._maxpool = mp
    ac = ac.to("cuda")
    feat = ac.encode(obs.to("cuda"))
    (feat * wfeat.to("cuda")).sum().backward()
    torch.save({"feat": feat.detach().cpu(), "pre_pool": rec, "grads": {k: p.grad.cpu() for k, p in ac.named_parameters() if p.grad is not None}}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    print("feat", rel(a["feat"], b["feat"]))
    for i, ((ya, ga), (yb, gb)) in enumerate(zip(a["pre_pool"], b["pre_pool"])):
        print(f"pool {i}: pre-pool rel diff {rel(ya, yb):.2e}, argmax entries differing {int((ga != gb).sum())} of {ga.numel()}")
    for k in a["grads"]:
        print(k, f"{rel(a['grads'][k], b['grads'][k]):.2e}")
