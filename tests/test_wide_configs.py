"""Networks WIDER than the reference's one published configuration (tests/wide_configs.py): the reference's constructors take any
`channels` / `depths` lists (models/blocks.py:183-222, rew_end_model.py:93-133, actor_critic.py:101-113), the kernels are
instantiated for the default configuration's shapes, and the host covers the rest by decomposition (engine._conv2d_wide: input
channels in runs of <= 256; ac_native._wgrad_tiled: the (Cout, Cin) plane in 64 x 64 tiles; ac_native.gn_bwd_sliced: whole
GroupNorm groups).  Held against fixtures produced by EXECUTING THE REFERENCE on the same configurations
(tests/golden/make_golden.py --wide -> tests/golden/wide.pt), at the bars of the default configuration's tests (1e-4).

Every test runs twice: the product's host code against the SIMT-interpreter build of the kernels (tests/simt: test infrastructure,
the CPU suite) and on the device (`-m gpu`, part of the round's GPU suite since round 6: the branches were written in a session without
a GPU and ran there for the first time in round 6).  The kernel-level test at the end checks the decomposition itself against fp64 torch."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests import wide_configs as W
from tests.simt.host_harness import engine_on_interpreter

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wide.pt")
# Every test below runs on the interpreter (the CPU suite) and on the device (`pytest -m gpu`).
BACKENDS = ["interpreter", pytest.param("cuda", marks=[pytest.mark.gpu])]


def _backend(kind):
    """(context manager under which the product's host code runs, device of the tensors)"""
    import contextlib

    return (engine_on_interpreter(), "cpu") if kind == "interpreter" else (contextlib.nullcontext(), "cuda")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


class _Count:
    """stands in for native.PROFILER: launches per C-ABI entry point"""

    def __init__(self):
        self.n = {}

    def annotate(self, key, flops, nbytes):
        pass

    def call(self, name, fn, args):
        self.n[name] = self.n.get(name, 0) + 1
        return fn(*args)


def _denoiser():
    import diamond_amd as D
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.testing import fill_module_

    den = D.Denoiser(D.DenoiserConfig(inner_model=InnerModelConfig(**W.DENOISER), sigma_data=0.5, sigma_offset_noise=0.3))
    fill_module_(den, W.WEIGHT_SEED)
    return den


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_denoiser_model_output_vs_reference_golden(gold, monkeypatch, backend):
    from diamond_amd import native as nv
    from diamond_amd.testing import synthetic_actions, synthetic_frames

    ctx, dev = _backend(backend)
    den = _denoiser().eval().to(dev)
    s = W.SIZE
    g = torch.Generator().manual_seed(5)
    obs, act, x = synthetic_frames(g, 2, 12, s, s).to(dev), synthetic_actions(g, 4, 2, 4).to(dev), torch.randn(2, 3, s, s, generator=g).to(dev)
    counter = _Count()
    monkeypatch.setattr(nv, "PROFILER", counter)
    with ctx, torch.no_grad():
        for i, sigma in enumerate((torch.tensor(0.7), torch.tensor([0.05, 3.0]))):
            f = den.compute_model_output(x, obs, act, sigma.to(dev)).cpu()
            err = rel(f, gold[f"model_output_{i}"])
            assert err < 1e-4, (i, err)
    # the wide convolutions really ran as chains: more launches than the network has convolutions (2 forwards)
    convs = sum(1 for m in den.modules() if isinstance(m, torch.nn.Conv2d))
    assert counter.n["dmd_conv2d"] > 2 * convs, (counter.n, convs)


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_denoiser_training_step_vs_reference_golden(gold, backend):
    """Denoiser.forward + loss.backward(): loss, every gradient tensor (sampled) and every gradient norm of the recorded forward
    + hand-written backward against the reference's autograd, split-fp16 and exact fp32 arithmetic"""
    from types import SimpleNamespace

    import diamond_amd as D
    from diamond_amd import unet_train as UT
    from diamond_amd.testing import synthetic_actions, synthetic_frames

    ctx, dev = _backend(backend)
    den = _denoiser().train().to(dev)
    den.setup_training(D.SigmaDistributionConfig(**W.SIGMA_DIST))
    s = W.SIZE
    g = torch.Generator().manual_seed(31)
    obs, act = synthetic_frames(g, 1, 5, 3, s, s).to(dev), synthetic_actions(g, 4, 1, 5).to(dev)
    batch = SimpleNamespace(obs=obs, act=act, mask_padding=torch.ones(1, 5, dtype=torch.bool).to(dev))
    den.randn_fn = lambda shape: torch.randn(*shape)  # CPU default generator: the stream the reference consumed
    ref = gold["train"]
    try:
        # (the exact-fp32 arithmetic as well where time allows: on the device, or with DIAMOND_SLOW_CPU_TESTS=1)
        both = backend != "interpreter" or os.environ.get("DIAMOND_SLOW_CPU_TESTS") == "1"
        for precision in ("f16x2", "f32") if both else ("f16x2",):
            UT.TRAIN_PRECISION = precision
            with _backend(backend)[0]:
                torch.manual_seed(77)
                den.zero_grad()
                loss, _ = den(batch)
                loss.backward()
            errs = {"loss": rel(loss.detach().cpu(), ref["loss"])}
            for k, p in den.named_parameters():
                assert p.grad is not None, f"no gradient for {k}"
                errs["grad " + k] = rel(W.sample_grad(p.grad).cpu(), ref["grads"][k])
                n = float(ref["grad_norms"][k])
                errs["|grad| " + k] = abs(float(p.grad.double().norm()) - n) / (n + 1e-30)
            bad = {k: v for k, v in errs.items() if v >= 1e-4}
            assert not bad, (precision, bad)
    finally:
        UT.TRAIN_PRECISION = "f16x2"


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_rew_end_model_and_actor_critic_vs_reference_golden(gold, backend):
    from diamond_amd.actor_critic import ActorCritic, ActorCriticConfig
    from diamond_amd.rew_end_model import RewEndModel, RewEndModelConfig
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    s = W.SIZE
    ctx, dev = _backend(backend)
    m = RewEndModel(RewEndModelConfig(**W.REW_END))
    fill_module_(m, W.WEIGHT_SEED + 1)
    m.eval().to(dev)
    g = torch.Generator().manual_seed(9)
    obs, act = synthetic_frames(g, 2, 3, 3, s, s).to(dev), synthetic_actions(g, 4, 2, 2).to(dev)
    with ctx, torch.no_grad():
        lr, le, (h, c) = [t.cpu() if torch.is_tensor(t) else tuple(u.cpu() for u in t) for t in m.predict_rew_end(obs[:, :-1], act, obs[:, 1:])]
    r = gold["rew_end"]
    errs = {"logits_rew": rel(lr, r["logits_rew"]), "logits_end": rel(le, r["logits_end"]), "h": rel(h, r["h"]), "c": rel(c, r["c"])}
    assert max(errs.values()) < 1e-4, errs

    ac = ActorCritic(ActorCriticConfig(**W.ACTOR_CRITIC))
    fill_module_(ac, W.WEIGHT_SEED + 2)
    ac.to(dev)
    g = torch.Generator().manual_seed(11)
    obs = synthetic_frames(g, 2, 3, s, s).to(dev)
    with _backend(backend)[0]:
        o = ac.predict_act_value(obs, None)
        (o.logits_act.square().sum() + o.val.sum()).backward()
    r = gold["actor_critic"]
    errs = {"logits": rel(o.logits_act.detach().cpu(), r["logits"]), "val": rel(o.val.detach().cpu(), r["val"])}
    for k, p in ac.named_parameters():
        errs["grad " + k] = rel(W.sample_grad(p.grad).cpu(), r["grads"][k])
        n = float(r["grad_norms"][k])
        errs["|grad| " + k] = abs(float(p.grad.double().norm()) - n) / (n + 1e-30)
    bad = {k: v for k, v in errs.items() if v >= 1e-4}
    assert not bad, bad


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_full_window_vs_reference_golden(monkeypatch, backend):
    """The whole north-star path on the wide networks: two BPTT windows of ActorCritic.forward() + backward through WorldModelEnv /
    env_loop / DiffusionSampler / the reward-end model with resets and burn-in, the reference's RNG order -- sampled actions,
    rewards, ends and truncations BIT-exact against the reference's own rollout (tests/golden/make_golden.py --wide-window)."""
    import diamond_amd as D
    from diamond_amd.actor_critic import ActorCriticConfig
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.rew_end_model import RewEndModelConfig
    from diamond_amd.testing import fill_module_
    from tests import test_gpu_models as M

    if backend == "interpreter" and os.environ.get("DIAMOND_SLOW_CPU_TESTS") != "1":
        pytest.skip("3-4 minutes on 8 cores: DIAMOND_SLOW_CPU_TESTS=1 runs it")
    ctx, dev = _backend(backend)
    monkeypatch.setattr(M, "DEV", dev)
    ag = D.Agent(W.agent_config(D.AgentConfig, D.DenoiserConfig, InnerModelConfig, RewEndModelConfig, ActorCriticConfig))
    fill_module_(ag, M.WEIGHT_SEED)
    with ctx:
        M.test_full_window_vs_reference_golden("window_wide.pt", ag.eval().to(dev))


@pytest.mark.parametrize("case", ["cat 320 + 64, 3x3", "one 512-channel source, 1x1", "stride 2", "upsample"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_convolution_decomposition_vs_fp64(case, backend):
    """engine._conv2d_wide on its own: GroupNorm(+affine)+SiLU prologue on a source wider than one launch takes, a second raw
    source, bias, residual, the emitted statistics of the finished output -- against fp64 torch ops on the same values"""
    from diamond_amd import engine as E
    from diamond_amd import native as nv
    from diamond_amd.engine import Act, NormSpec

    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 8, 8
    c_a, c_b, cout, taps, stride, upsample = {"cat 320 + 64, 3x3": (320, 64, 96, 9, 1, False), "one 512-channel source, 1x1": (512, 0, 64, 1, 1, False),
                                              "stride 2": (288, 32, 64, 9, 2, False), "upsample": (320, 0, 32, 9, 1, True)}[case]
    hs, ws = (h * 2, w * 2) if stride == 2 else ((h // 2, w // 2) if upsample else (h, w))
    a = torch.randn(n, hs, ws, c_a, generator=g)
    b = torch.randn(n, hs, ws, c_b, generator=g) if c_b else None
    gamma, beta = torch.randn(c_a, generator=g) * 0.3 + 1, torch.randn(c_a, generator=g) * 0.2
    k = 3 if taps == 9 else 1
    wt = torch.randn(cout, c_a + c_b, k, k, generator=g) / (taps * (c_a + c_b)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, h, w, cout, generator=g)
    prologue = nv.PROLOGUE_NONE if upsample else nv.PROLOGUE_NORM_SILU  # (the kernels fold no normalisation into an upsampling gather)
    # fp64 reference
    xa = a.double().permute(0, 3, 1, 2)
    if prologue != nv.PROLOGUE_NONE:
        xa = F.silu(F.group_norm(xa, c_a // 32, gamma.double(), beta.double(), eps=1e-5))
    xin = xa if b is None else torch.cat([xa, b.double().permute(0, 3, 1, 2)], 1)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wt.double(), bias.double(), stride=stride, padding=k // 2) + res.double().permute(0, 3, 1, 2)
    ctx, dev = _backend(backend)
    on = lambda t: t.contiguous().to(dev)
    with ctx:
        sa = E.gn_stats(on(a)) if prologue != nv.PROLOGUE_NONE else Act(on(a))
        srcs = [(sa, prologue, NormSpec(mul=on(gamma), add=on(beta)) if prologue != nv.PROLOGUE_NONE else None)]
        if b is not None:
            srcs.append((Act(on(b)), nv.PROLOGUE_NONE, None))
        out = E.conv2d(srcs, nv.pack_conv_weight(on(wt)), on(bias), cout, taps=taps, stride=stride, upsample=upsample, residual=Act(on(res)))
    got = out.t.cpu().double().permute(0, 3, 1, 2)
    assert rel(got, ref) < 2e-6, rel(got, ref)
    # the partial sums the last launch emitted describe the finished output
    sums = out.stats.cpu().sum(2)  # (n, groups, 2)
    want = ref.reshape(n, cout // 32, 32 * h * w)
    assert rel(sums[..., 0], want.sum(-1)) < 1e-5 and rel(sums[..., 1], want.square().sum(-1)) < 1e-5
