"""One-off GPU diagnostics (round 2): (1) which GroupNorm statistics slots differ between a tiles_per_wg >= 2 launch
and the N=2 launch of the same images (32-cout instance); (2) is torch's (rocBLAS/hipBLASLt) fp32 GEMM exact enough
for the LSTM weight gradient shape; (3) actor-critic encoder gradients at B=256 against fp64 and fp32 CPU oracles."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E, native as nv

DEV = "cuda"
what = sys.argv[1:] or ["stats", "gemm", "acgrad"]

if "stats" in what:
    for (n, h, cin, cout, res_on) in ((256, 64, 32, 32, True), (511, 32, 64, 32, False), (64, 64, 32, 32, False), (48, 64, 32, 32, False)):
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(n, h, h, cin, device=DEV, generator=g) * 1.5 + 0.3
        wgt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / math.sqrt(cin * 9)
        bias = torch.randn(cout, device=DEV, generator=g) * 0.1
        res = torch.randn(n, h, h, cout, device=DEV, generator=g) if res_on else None
        wp, w16, bp = nv.pack_conv_weight(wgt), nv.pack_conv_weight_f16x2(wgt), nv.pad_vector(bias, 32)

        def run(xx, rr):
            return E.conv2d([(E.Act(xx), 0, None)], wp, bp, cout, residual=None if rr is None else E.Act(rr), w_f16=w16)

        big = run(x, res)
        torch.cuda.synchronize()
        nbad_img, shown = 0, 0
        for i in range(0, n, 2):
            pi = torch.tensor([i, min(i + 1, n - 1)], device=DEV)
            small = run(x[pi].contiguous(), None if res is None else res[pi].contiguous())
            eq_out = torch.equal(big.t[pi], small.t)
            d = (big.stats[pi] != small.stats).any(-1)  # (2, G, T)
            if d.any() or not eq_out:
                nbad_img += 1
                if shown < 6:
                    shown += 1
                    idx = d.nonzero()
                    print(f"N={n} {h}x{h} cin{cin}: images {i},{i+1}: outputs equal {eq_out}; differing stat slots (img, g, t): {idx.tolist()[:12]}")
                    for (a, gg, t) in idx.tolist()[:4]:
                        bs, ss = big.stats[pi[a], gg, t].tolist(), small.stats[a, gg, t].tolist()
                        print(f"    slot t={t}: big {bs}  small {ss}  diff {[b_ - s_ for b_, s_ in zip(bs, ss)]}")
        tiles = (n * (h // 16) ** 2 + 1) // 2
        print(f"N={n} {h}x{h} cin{cin} cout{cout}: tiles {tiles} tpw {(tiles + 255) // 256}: {nbad_img} image pairs differ", flush=True)

if "gemm" in what:
    g = torch.Generator().manual_seed(0)
    for (b, m, k) in ((3, 2048, 1024), (4, 2048, 1024), (256, 2048, 1024), (256, 2048, 512), (3, 4, 512)):
        dg = torch.randn(b, m, generator=g) * 0.01
        xx = torch.randn(b, k, generator=g).abs() * 2
        ref = dg.double().t() @ xx.double()
        got = (dg.to(DEV).t() @ xx.to(DEV)).cpu().double()
        cpu = (dg.t() @ xx).double()
        rel = lambda a: float((a - ref).abs().max() / ref.abs().max())
        nrm = lambda a: float(abs(a.norm() - ref.norm()) / ref.norm())
        print(f"dW = dg^T x, B={b} ({m}x{k}): torch-GPU max-rel {rel(got):.2e} norm-rel {nrm(got):.2e} | torch-CPU {rel(cpu):.2e} {nrm(cpu):.2e}")
        y = torch.nn.functional.linear(xx.to(DEV), torch.randn(m, k, generator=g).to(DEV) / 32)
    print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "fp32 precision", torch.get_float32_matmul_precision(), flush=True)

if "acgrad" in what:
    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_frames
    from oracle import diamond_oracle as O

    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, 5)
    ac = agent.actor_critic
    g = torch.Generator().manual_seed(258)
    b = 256
    obs = synthetic_frames(g, b, 3, 64, 64)
    wfeat = torch.randn(b, 1024, generator=g) / (b * 15)
    grads = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in ac.state_dict().items()}
        t0 = time.time()
        ref = O.ac_encoder(sd, O.ActorCriticSpec(), obs.to(dt)).flatten(1)
        (ref * wfeat.to(dt)).sum().backward()
        grads[dt] = {k: v.grad for k, v in sd.items() if v.grad is not None}
        print(f"oracle {dt}: {time.time() - t0:.1f}s", flush=True)
    ac = ac.to(DEV)
    feat = ac.encode(obs.to(DEV))
    (feat * wfeat.to(DEV)).sum().backward()
    rel = lambda a, r: float((a.double().cpu() - r.double()).abs().max() / r.double().abs().max())
    for k, p in ac.named_parameters():
        if k.startswith("encoder."):
            print(f"{k:40s} hip-vs-fp64 {rel(p.grad, grads[torch.float64][k]):.2e}  cpu32-vs-fp64 {rel(grads[torch.float32][k], grads[torch.float64][k]):.2e}  "
                  f"hip-vs-cpu32 {rel(p.grad, grads[torch.float32][k]):.2e}")
