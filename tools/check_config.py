"""Does a network CONFIGURATION run correctly through diamond_amd?  Builds the reference's module and this package's module on
the same configuration and the same (name-keyed) weights, runs both on the same inputs and prints the relative errors.

    python tools/check_config.py denoiser       --channels 64,128,256,256 --depths 1,2,2,1 --attn 0,0,1,1 --size 64
    python tools/check_config.py denoiser-train --channels 96,96,160,320 --depths 1,1,1,1 --attn 0,0,0,0 --size 64
    python tools/check_config.py rew-end        --channels 64,96,320 --depths 1,1,1 --attn 0,0,1 --size 32 [--train]
    python tools/check_config.py actor-critic   --channels 32,64,96,160 --down 1,1,1,1 --size 32
    python tools/check_config.py sampler        --steps 4 --order 2 --s-churn 1.0 --s-tmax 50

Build container only: it imports /root/reference (tests/golden/_refimport.py) and, without a GPU, runs the product's host code
against the SIMT-interpreter build of the kernels (tests/simt: test infrastructure -- correctness of the arithmetic and of the host
orchestration, nothing about speed or the hardware's memory ordering).  With a GPU (`--device cuda`) the product runs as shipped.
This is how the configurations listed in HISTORY.md section 4 ("Configurations other than the published one") were probed."""
import argparse
import contextlib
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def ints(s):
    return [int(v) for v in s.split(",")]


def rel(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))


def worst(mine, ref, k=3):
    errs = {n: rel(p.grad, dict(ref.named_parameters())[n].grad) for n, p in mine.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:k]
    return max(errs.values()), [(n, f"{v:.1e}") for n, v in top]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("what", choices=("denoiser", "denoiser-train", "rew-end", "actor-critic", "sampler"))
    ap.add_argument("--channels", type=ints, default=None)
    ap.add_argument("--depths", type=ints, default=None)
    ap.add_argument("--attn", type=ints, default=None)
    ap.add_argument("--down", type=ints, default=[1, 1, 1, 1])
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--cond-channels", type=int, default=None)
    ap.add_argument("--num-actions", type=int, default=4)
    ap.add_argument("--train", action="store_true", help="rew-end: the training step (RewEndModel.forward + backward) instead of predict_rew_end")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--s-churn", type=float, default=0.0)
    ap.add_argument("--s-tmin", type=float, default=0.0)
    ap.add_argument("--s-tmax", type=float, default=float("inf"))
    ap.add_argument("--s-noise", type=float, default=1.0)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()

    import _refimport as R

    assert R.available(), "the reference (/root/reference) is needed: build container only"
    R.install()
    import diamond_amd as D
    from diamond_amd.actor_critic import ActorCritic, ActorCriticConfig
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.rew_end_model import RewEndModel, RewEndModelConfig
    from diamond_amd.testing import fill_module_, rew_end_train_batch, synthetic_actions, synthetic_frames

    dev = args.device
    if dev == "cpu":
        from tests.simt.host_harness import engine_on_interpreter as backend
    else:
        backend = contextlib.nullcontext
    on = lambda t: t.to(dev)
    s, b, na = args.size, args.batch, args.num_actions
    g = torch.Generator().manual_seed(5)

    if args.what in ("denoiser", "denoiser-train", "sampler"):
        from data import Batch
        from models.diffusion import Denoiser as RDen, DenoiserConfig as RDC, DiffusionSampler as RS, DiffusionSamplerConfig as RSC, \
            InnerModelConfig as RIC, SigmaDistributionConfig as RSD

        kw = dict(img_channels=3, num_steps_conditioning=4, cond_channels=args.cond_channels or 256, depths=args.depths or [2, 2, 2, 2],
                  channels=args.channels or [64, 64, 64, 64], attn_depths=args.attn or [0, 0, 0, 0], num_actions=na)
        den = D.Denoiser(D.DenoiserConfig(inner_model=InnerModelConfig(**kw), sigma_data=0.5, sigma_offset_noise=0.3))
        fill_module_(den, 3)
        ref = RDen(RDC(inner_model=RIC(**kw), sigma_data=0.5, sigma_offset_noise=0.3))
        ref.load_state_dict(den.state_dict(), strict=True)
        den.to(dev)
        if args.what == "denoiser":
            den.eval(), ref.eval()
            obs, act, x = synthetic_frames(g, b, 12, s, s), synthetic_actions(g, na, b, 4), torch.randn(b, 3, s, s, generator=g)
            for sigma in (torch.tensor(0.7), torch.linspace(0.05, 3.0, b)):
                with torch.no_grad():
                    f_ref = ref.compute_model_output(x, obs, act, ref.compute_conditioners(sigma))
                    with backend():
                        f = den.compute_model_output(on(x), on(obs), on(act), on(sigma))
                print(f"model output, sigma {tuple(sigma.shape) or 'scalar'}: rel err {rel(f, f_ref):.2e}")
        elif args.what == "denoiser-train":
            sd = dict(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20)
            den.train(), ref.train()
            den.setup_training(D.SigmaDistributionConfig(**sd)), ref.setup_training(RSD(**sd))
            obs, act = synthetic_frames(g, b, 5, 3, s, s), synthetic_actions(g, na, b, 5)
            mask = torch.ones(b, 5, dtype=torch.bool)
            torch.manual_seed(77)
            loss_r, _ = ref(Batch(obs=obs, act=act, rew=None, end=None, trunc=None, mask_padding=mask, info=None, segment_ids=None))
            loss_r.backward()
            den.randn_fn = lambda shape: torch.randn(*shape).to(dev)  # the CPU stream the reference consumed
            with backend():
                torch.manual_seed(77)
                loss, _ = den(SimpleNamespace(obs=on(obs), act=on(act), mask_padding=on(mask)))
                loss.backward()
            w, top = worst(den, ref)
            print(f"training step: loss rel err {rel(loss, loss_r):.2e}; worst gradient rel err {w:.2e} {top}")
        else:
            den.eval(), ref.eval()
            cfg = dict(num_steps_denoising=args.steps, order=args.order, s_churn=args.s_churn, s_tmin=args.s_tmin, s_tmax=args.s_tmax, s_noise=args.s_noise)
            obs, act = synthetic_frames(g, b, 4, 3, s, s), synthetic_actions(g, na, b, 4)
            torch.manual_seed(99)
            with torch.no_grad():
                xr, trr = RS(ref, RSC(**cfg)).sample(obs, act)
            sm = D.DiffusionSampler(den, D.DiffusionSamplerConfig(**cfg))
            sm.noise_fn = lambda shape, d: torch.randn(*shape).to(d)
            torch.manual_seed(99)
            with backend(), torch.no_grad():
                x, tr = sm.sample(on(obs), on(act))
            lvl = lambda t: t.cpu().clamp(-1, 1).add(1).div(2).mul(255).round()
            for k, (a, r) in enumerate(zip(tr, trr)):
                print(f"trajectory[{k}]: max abs diff {float((a.cpu() - r).abs().max()):.3g}, values differing {float(((a.cpu() - r).abs() > 1e-6).float().mean()):.2e}")
            print(f"sampled frame: pixels on another uint8 level {float((lvl(x) != lvl(xr)).float().mean()):.2e} "
                  "(free-running: one flipped level of a denoised frame feeds the next evaluation; Heun divides it by sigma_next)")
    elif args.what == "rew-end":
        from data import Batch
        from models.rew_end_model import RewEndModel as RRE, RewEndModelConfig as RREC

        kw = dict(lstm_dim=512, img_channels=3, img_size=s, cond_channels=args.cond_channels or 128, depths=args.depths or [2, 2, 2, 2],
                  channels=args.channels or [32, 32, 32, 32], attn_depths=args.attn or [0, 0, 0, 0], num_actions=na)
        m = RewEndModel(RewEndModelConfig(**kw))
        fill_module_(m, 4)
        r = RRE(RREC(**kw))
        r.load_state_dict(m.state_dict(), strict=True)
        m.to(dev)
        if args.train:
            assert s == 64, "the synthetic training segment is 64 x 64"
            m.train(), r.train()
            bd = rew_end_train_batch(torch.Generator().manual_seed(41))
            loss_r, logs_r = r(Batch(**bd))
            loss_r.backward()
            with backend():
                loss, logs = m(SimpleNamespace(**{k: (on(v) if torch.is_tensor(v) else v) for k, v in bd.items()}))
                loss.backward()
            w, top = worst(m, r)
            same = all(torch.equal(logs["confusion_matrix"][k].cpu(), logs_r["confusion_matrix"][k]) for k in ("rew", "end"))
            print(f"training step: loss rel err {rel(loss, loss_r):.2e}; worst gradient rel err {w:.2e} {top}; confusion matrices equal: {same}")
        else:
            m.eval(), r.eval()
            obs, act = synthetic_frames(g, b, 3, 3, s, s), synthetic_actions(g, na, b, 2)
            with torch.no_grad():
                lr_, le_, (h_, c_) = r.predict_rew_end(obs[:, :-1], act, obs[:, 1:])
                with backend():
                    lr, le, (h, c) = m.predict_rew_end(on(obs[:, :-1]), on(act), on(obs[:, 1:]))
            print(f"predict_rew_end: logits_rew {rel(lr, lr_):.2e}, logits_end {rel(le, le_):.2e}, h {rel(h, h_):.2e}, c {rel(c, c_):.2e}")
    else:
        from models.actor_critic import ActorCritic as RAC, ActorCriticConfig as RACC

        kw = dict(lstm_dim=512, img_channels=3, img_size=s, channels=args.channels or [32, 32, 64, 64], down=args.down, num_actions=na)
        ac = ActorCritic(ActorCriticConfig(**kw))
        fill_module_(ac, 5)
        r = RAC(RACC(**kw))
        r.load_state_dict(ac.state_dict(), strict=True)
        ac.to(dev)
        obs = synthetic_frames(g, b, 3, s, s)
        o_r = r.predict_act_value(obs, None)
        (o_r.logits_act.square().sum() + o_r.val.sum()).backward()
        with backend():
            o = ac.predict_act_value(on(obs), None)
            (o.logits_act.square().sum() + o.val.sum()).backward()
        w, top = worst(ac, r)
        print(f"predict_act_value: logits {rel(o.logits_act, o_r.logits_act):.2e}, value {rel(o.val, o_r.val):.2e}; worst gradient rel err {w:.2e} {top}")


if __name__ == "__main__":
    main()
