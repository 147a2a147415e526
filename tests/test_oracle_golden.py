"""Pin the CPU oracle (oracle/diamond_oracle.py) against fixtures produced by executing the
reference itself (tests/golden/make_golden.py).  CPU only; runs everywhere."""
import os

import pytest
import torch

from oracle import diamond_oracle as O
from tests.conftest import load_golden, make_oracle_agent
from diamond_amd.testing import initial_condition_batches, synthetic_actions, synthetic_frames

torch.set_num_threads(8)


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def u8(x):
    return x.add(1).div(2).mul(255).round().to(torch.uint8)


def check_quantised(mine_u8, ref_u8, max_frac=1e-4):
    diff = (mine_u8.int() - ref_u8.int()).abs()
    assert int(diff.max()) <= 1, "a pixel differs by more than one uint8 level"
    assert float((diff > 0).float().mean()) <= max_frac


def _denoiser_case(tag, attn_depths, b, h=64, w=64):
    gold = load_golden(f"denoiser_{tag}.pt")
    a = make_oracle_agent(attn_depths=attn_depths)
    g = torch.Generator().manual_seed(gold["seed"])
    obs = synthetic_frames(g, b, 12, h, w)
    act = synthetic_actions(g, 4, b, 4)
    noise = torch.randn(b, 3, h, w, generator=g)
    sig = O.build_sigmas(a.sspec)
    assert torch.equal(sig, gold["sigmas"])
    for i, sigma in enumerate(list(sig[:-1]) + [torch.tensor([0.7, 1.9][:b])]):
        if f"model_output_{i}" not in gold:
            continue
        x = noise * sigma.reshape(-1, 1, 1, 1) + obs[:, -3:] * 0.5
        d, f = O.denoise(a.denoiser, a.dspec, x, sigma, obs, act, return_model_output=True)
        assert rel_err(f, gold[f"model_output_{i}"]) < 2e-5, (tag, i)
        check_quantised(u8(d), gold[f"denoised_u8_{i}"], max_frac=2e-4)
        if i == 0:
            c = O.cond_vector(a.denoiser, O.conditioners(a.dspec, sigma)[3], act)
            assert rel_err(c, gold["cond_0"]) < 1e-5


def test_denoiser_default():
    _denoiser_case("default", (0, 0, 0, 0), 2)


def test_denoiser_with_unet_attention():
    _denoiser_case("attn0011", (0, 0, 1, 1), 1)


def test_denoiser_sizes_off_the_tile_grid():
    """72x72 (U-Net levels 72 / 36 / 18 / 9, nothing padded by the reference) and 68x76 with attention (padded to 72x80 inside
    UNet.forward and cropped back, /root/reference/src/models/blocks.py:227-229,247)."""
    _denoiser_case("72x72", (0, 0, 0, 0), 2, 72, 72)
    _denoiser_case("attn0011_68x76", (0, 0, 1, 1), 1, 68, 76)


@pytest.mark.parametrize("tag,attn", [("default", (0, 0, 0, 0)), ("72x72", (0, 0, 0, 0))])
def test_quantised_frame_budget_on_200k_pixels(tag, attn):
    """the oracle against the >= 200k-pixel fixtures (make_golden.py --pixels): the fraction of pixels on another uint8 level than
    the reference's, where 1e-4 is 20+ pixels (first sampler sigma and the per-sample sigmas; the GPU test does all of them)"""
    gold = load_golden(f"denoiser_pixels_{tag}.pt")
    b, h, w = gold["b"], gold["h"], gold["w"]
    a = make_oracle_agent(attn_depths=attn)
    g = torch.Generator().manual_seed(gold["seed"])
    obs = synthetic_frames(g, b, 12, h, w)
    act = synthetic_actions(g, 4, b, 4)
    noise = torch.randn(b, 3, h, w, generator=g)
    n = sum(1 for k in gold if k.startswith("denoised_u8_"))
    for i, sigma in ((0, gold["sigmas"][0]), (n - 1, gold["per_sample_sigma"])):
        x = noise * sigma.reshape(-1, 1, 1, 1) + obs[:, -3:] * 0.5
        d = O.denoise(a.denoiser, a.dspec, x, sigma, obs, act)
        check_quantised(u8(d), gold[f"denoised_u8_{i}"], max_frac=1e-4)


def test_sampler_euler_and_heun():
    gold = load_golden("sampler.pt")
    a = make_oracle_agent()
    g = torch.Generator().manual_seed(gold["seed"])
    for name, sspec, b in (("euler3", O.SamplerSpec(num_steps_denoising=3), 2),
                           ("heun4", O.SamplerSpec(num_steps_denoising=4, order=2), 1)):
        prev_obs = synthetic_frames(g, b, 4, 3, 64, 64)
        prev_act = synthetic_actions(g, 4, b, 4)
        torch.manual_seed(gold[name]["noise_seed"])
        noise = torch.randn(b, 3, 64, 64)
        x, traj = O.sample(a.denoiser, a.dspec, sspec, prev_obs, prev_act, noise)
        assert torch.equal(O.build_sigmas(sspec), gold[name]["sigmas"])
        traj = torch.stack(traj, 1)
        # free-running: a one-level quantisation flip early on propagates, so compare loosely
        # in value but require nearly all pixels of the final frame to sit on the same level
        assert (traj - gold[name]["trajectory"]).abs().max() < 2.5 * 2 / 255 * 5.0 / 0.28
        check_quantised(u8(x.clamp(-1, 1)), u8(gold[name]["x"].clamp(-1, 1)), max_frac=1e-3)


def heun_step_budget(sigmas, i):
    """A pixel of a quantised denoiser output on the neighbouring uint8 level (2/255) moves the Heun step's result by
    2/255 * |dt| / (2 sigma) through the first evaluation and 2/255 * |dt| / (2 sigma_next) through the second
    (reference diffusion_sampler.py:52-56); the last step (sigma_next = 0) is an Euler step."""
    s0, s1 = float(sigmas[i]), float(sigmas[i + 1])
    return 2 / 255 * (abs(s1 - s0) / s0 if s1 == 0 else abs(s1 - s0) * (0.5 / s0 + 0.5 / s1))


def test_sampler_heun5_every_step_teacher_forced():
    """BASELINE configs[3]'s sampler form at a batch of 2: each of the 5 Heun steps (9 denoiser calls) starts from the
    REFERENCE's own trajectory point, so quantisation flips cannot accumulate."""
    gold = load_golden("sampler_heun5.pt")
    a = make_oracle_agent()
    g = torch.Generator().manual_seed(gold["seed"])
    prev_obs = synthetic_frames(g, 2, 4, 3, 64, 64)
    prev_act = synthetic_actions(g, 4, 2, 4)
    sspec = O.SamplerSpec(num_steps_denoising=5, order=2)
    sig = O.build_sigmas(sspec)
    assert torch.equal(sig, gold["sigmas"])
    ref = gold["trajectory"]
    torch.manual_seed(gold["noise_seed"])
    assert torch.equal(torch.randn(2, 3, 64, 64), ref[:, 0])
    for i in range(5):
        x, _ = O.sample(a.denoiser, a.dspec, sspec, prev_obs, prev_act, ref[:, i], sigmas=sig[i:i + 2])
        diff = (x - ref[:, i + 1]).abs()
        assert float(diff.max()) <= heun_step_budget(sig, i) * 2 + 1e-4
        assert float((diff > 1e-4).float().mean()) <= 2e-4, (i, float((diff > 1e-4).float().mean()))


def test_sampler_heun_step_budget_on_200k_pixels():
    """the oracle's Heun step against the >= 200k-value fixture (make_golden.py --heun-pixels), one teacher-forced step"""
    gold = load_golden("sampler_heun_pixels.pt")
    b, sig = gold["b"], gold["sigmas"]
    a = make_oracle_agent()
    g = torch.Generator().manual_seed(gold["seed"])
    prev_obs = synthetic_frames(g, b, 4, 3, 64, 64)
    prev_act = synthetic_actions(g, 4, b, 4)
    i = gold["steps"][-1]
    spec = O.SamplerSpec(num_steps_denoising=5, order=2)
    x, _ = O.sample(a.denoiser, a.dspec, spec, prev_obs, prev_act, gold[f"x_{i}"], sigmas=sig[i:i + 2])
    diff = (x - gold[f"x_{i + 1}"]).abs()
    assert float(diff.max()) <= heun_step_budget(sig, i) * 2 + 1e-4
    assert float((diff > 1e-4).float().mean()) <= 2e-4


def test_rew_end_model():
    gold = load_golden("rew_end.pt")
    a = make_oracle_agent()
    g = torch.Generator().manual_seed(gold["seed"])
    obs = synthetic_frames(g, 2, 4, 3, 64, 64)
    act = synthetic_actions(g, 4, 2, 4)
    lr, le, (hx, cx) = O.rew_end_predict(a.rew_end_model, a.rspec, obs[:, :-1], act[:, :-1], obs[:, 1:])
    lr2, le2, (hx2, cx2) = O.rew_end_predict(a.rew_end_model, a.rspec, obs[:, -1:], act[:, -1:], obs[:, :1], (hx, cx))
    for mine, key in ((lr, "logits_rew"), (le, "logits_end"), (hx, "hx"), (cx, "cx"), (lr2, "logits_rew_step"),
                      (le2, "logits_end_step"), (hx2, "hx_step"), (cx2, "cx_step")):
        assert mine.shape == gold[key].shape
        assert rel_err(mine, gold[key]) < 2e-5, key


def test_actor_critic_forward_backward():
    gold = load_golden("actor_critic.pt")
    a = make_oracle_agent()
    sd = {k: v.clone().requires_grad_(True) for k, v in a.actor_critic.items()}
    g = torch.Generator().manual_seed(gold["seed"])
    b = 3
    obs = synthetic_frames(g, b, 3, 64, 64)
    obs2 = synthetic_frames(g, b, 3, 64, 64)
    hx = torch.randn(b, 512, generator=g) * 0.3
    cx = torch.randn(b, 512, generator=g) * 0.3
    l1, v1, (h1, c1) = O.ac_predict(sd, a.aspec, obs, hx, cx)
    l2, v2, (h2, c2) = O.ac_predict(sd, a.aspec, obs2, h1, c1)
    w = torch.randn(b, 4, generator=g)
    loss = (l2 * w).sum() + v2.square().sum() + v1.sum() + 0.1 * c2.sum()
    loss.backward()
    for mine, key in ((l1, "logits1"), (v1, "val1"), (l2, "logits2"), (v2, "val2"), (h2, "hx2"), (c2, "cx2")):
        assert rel_err(mine.detach(), gold[key]) < 2e-5, key
    assert rel_err(loss.detach(), gold["loss"]) < 2e-5
    for k, n in gold["grad_norms"].items():
        assert abs(float(sd[k].grad.double().norm()) - float(n)) <= 5e-5 * float(n) + 1e-7, k
    for k, gr in gold["grads_small"].items():
        assert rel_err(sd[k].grad, gr) < 5e-5, k


def test_categorical_sampling_matches_torch_multinomial():
    """Categorical(logits).sample() == argmax(softmax/E), E from the default generator."""
    from torch.distributions.categorical import Categorical

    logits = torch.randn(64, 5, generator=torch.Generator().manual_seed(3)) * 2
    torch.manual_seed(99)
    ref = Categorical(logits=logits).sample()
    torch.manual_seed(99)
    e = torch.empty(64, 5).exponential_(1)
    assert torch.equal(O.categorical_sample(logits, e), ref)
    logits3 = logits[:, None, :3].contiguous()
    torch.manual_seed(5)
    ref3 = Categorical(logits=logits3).sample()
    torch.manual_seed(5)
    e3 = torch.empty(64, 1, 3).exponential_(1)
    assert torch.equal(O.categorical_sample(logits3, e3), ref3)


def test_full_window_rollout_and_loss():
    """Two BPTT windows through the oracle env + rollout driver vs the reference's own
    WorldModelEnv/env_loop/ActorCritic: bit-exact integer trajectories (act/end/trunc/rew),
    frames on the same uint8 levels, loss and gradient norms within fp32 noise."""
    gold = load_golden("window.pt")
    a = make_oracle_agent()
    a.actor_critic = {k: v.clone().requires_grad_(True) for k, v in a.actor_critic.items()}
    b, t = gold["b"], gold["backup_every"]
    draws = O.DrawSource(torch.Generator().manual_seed(gold["rng_seed"]))
    torch.manual_seed(gold["rng_seed"])
    draws.g = torch.default_generator  # same stream the reference consumed
    env = O.ImaginationEnv(a, initial_condition_batches(gold["pool_seed"], b, 4), b, gold["horizon"], draws,
                           num_batches_to_preload=gold["preload"])
    lc = O.LossSpec(backup_every=t)
    state = (env.reset(), torch.zeros(b, 512), torch.zeros(b, 512))
    for w in gold["windows"]:
        for p in a.actor_critic.values():
            p.grad = None
        (obs, act, rew, end, trunc, logits, val, vb), state = O.rollout(a, env, state, t, draws)
        assert torch.equal(act, w["act"]) and torch.equal(end, w["end"]) and torch.equal(trunc, w["trunc"])
        assert torch.equal(rew, w["rew"])
        check_quantised(u8(obs), w["obs_u8"], max_frac=1e-3)
        # free-running: the frames carry rare one-level uint8 flips (denoiser.py:83), each of
        # which moves the actor-critic outputs by O(1e-3) -- tight tolerances need teacher
        # forcing (done per component above); here the bound is the quantisation noise.
        assert rel_err(logits.detach(), w["logits_act"]) < 1e-2
        assert rel_err(val.detach(), w["val"]) < 1e-2
        assert rel_err(vb, w["val_bootstrap"]) < 1e-2
        loss, metrics = O.ac_loss(logits, val, act, rew, end, trunc, vb, lc)
        assert rel_err(loss.detach(), w["loss"]) < 1e-2
        loss.backward()
        for k, n in w["grad_norms"].items():
            assert abs(float(a.actor_critic[k].grad.norm()) - float(n)) <= 2e-2 * float(n) + 1e-6, k


def test_reference_bytecode_runs_a_window():
    """oracle/_ref (the reference's own modules as bytecode, oracle/make_ref.py) -- bench.py's `cpu_baseline.kind == "reference"`
    leg: one tiny window of ActorCritic.forward() + backward through the reference's WorldModelEnv / env_loop, in a child process
    (the stubs it installs must not leak into this one)."""
    import json
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import reference_window as RW
    finally:
        sys.path.pop(0)
    where, what = RW.reference_location()
    if where is None:
        pytest.skip(what)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "reference_window.py"), "--threads", "4", "--batch", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp", timeout=600,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HIP_VISIBLE_DEVICES=""))
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert d["kind"] == "reference" and d["value"] > 0 and d["cores"] == 4 and "15 imagined steps" in d["sample"]


def test_oracle_on_configurations_wider_than_the_default_one():
    """The oracle is spec-driven (depths / channels / attention per level): pinned here against the reference on the wide
    configurations of tests/wide_configs.py (tests/golden/wide.pt, make_golden.py --wide) -- denoiser model output at a scalar and
    a per-sample sigma, reward / end logits and LSTM state, actor-critic logits and values."""
    import diamond_amd as D
    from diamond_amd.actor_critic import ActorCritic, ActorCriticConfig
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.rew_end_model import RewEndModel, RewEndModelConfig
    from diamond_amd.testing import fill_module_
    from tests import wide_configs as W

    gold = load_golden("wide.pt")
    s = W.SIZE
    pick = lambda cfg, *ks: {k: tuple(cfg[k]) if isinstance(cfg[k], list) else cfg[k] for k in ks}
    # (the package's modules only as parameter containers: names and shapes of the state dict, filled by name)
    den = D.Denoiser(D.DenoiserConfig(inner_model=InnerModelConfig(**W.DENOISER), sigma_data=0.5, sigma_offset_noise=0.3))
    fill_module_(den, W.WEIGHT_SEED)
    spec = O.DenoiserSpec(**pick(W.DENOISER, "img_channels", "num_steps_conditioning", "cond_channels", "depths", "channels", "attn_depths"))
    g = torch.Generator().manual_seed(5)
    obs, act, x = synthetic_frames(g, 2, 12, s, s), synthetic_actions(g, 4, 2, 4), torch.randn(2, 3, s, s, generator=g)
    for i, sigma in enumerate((torch.tensor(0.7), torch.tensor([0.05, 3.0]))):
        f = O.model_output(dict(den.state_dict()), spec, x, sigma, obs, act)
        assert rel_err(f, gold[f"model_output_{i}"]) < 2e-5, i

    m = RewEndModel(RewEndModelConfig(**W.REW_END))
    fill_module_(m, W.WEIGHT_SEED + 1)
    rspec = O.RewEndSpec(**pick(W.REW_END, "lstm_dim", "img_channels", "img_size", "cond_channels", "depths", "channels", "attn_depths"))
    g = torch.Generator().manual_seed(9)
    obs, act = synthetic_frames(g, 2, 3, 3, s, s), synthetic_actions(g, 4, 2, 2)
    lr, le, (h, c) = O.rew_end_predict(dict(m.state_dict()), rspec, obs[:, :-1], act, obs[:, 1:])
    r = gold["rew_end"]
    assert max(rel_err(lr, r["logits_rew"]), rel_err(le, r["logits_end"]), rel_err(h, r["h"]), rel_err(c, r["c"])) < 2e-5

    ac = ActorCritic(ActorCriticConfig(**W.ACTOR_CRITIC))
    fill_module_(ac, W.WEIGHT_SEED + 2)
    aspec = O.ActorCriticSpec(**pick(W.ACTOR_CRITIC, "lstm_dim", "img_channels", "img_size", "channels", "down"))
    g = torch.Generator().manual_seed(11)
    obs = synthetic_frames(g, 2, 3, s, s)
    z = torch.zeros(2, aspec.lstm_dim)
    logits, val, _ = O.ac_predict({k: v.detach() for k, v in ac.state_dict().items()}, aspec, obs, z, z)
    r = gold["actor_critic"]
    assert max(rel_err(logits, r["logits"]), rel_err(val, r["val"])) < 2e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference (build container only)")
def test_committed_fixtures_regenerate_from_the_reference(tmp_path, monkeypatch):
    """The committed fixtures ARE what the reference computes here and now: a subset of them (the cheap ones: seconds) regenerated
    by the committed generator into a scratch directory and compared tensor by tensor with tests/golden/.  (All of them were
    regenerated and compared this way at the end of round 5: 0 differing tensors, HISTORY.md section 6.)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    committed = mg.HERE
    monkeypatch.setattr(mg, "HERE", str(tmp_path))
    agent = mg.ref_agent()
    names = ["rew_end.pt", "actor_critic.pt"]
    mg.gen_rew_end(agent)
    mg.gen_actor_critic(agent)
    if os.environ.get("DIAMOND_SLOW_CPU_TESTS") == "1":  # (another minute)
        mg.gen_denoiser(agent, "default")
        mg.gen_wide()
        names += ["denoiser_default.pt", "wide.pt"]

    def flat(o, pre=""):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from flat(v, f"{pre}/{k}")
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                yield from flat(v, f"{pre}/{i}")
        else:
            yield pre, o

    for name in names:
        a = dict(flat(torch.load(os.path.join(committed, name), weights_only=False)))
        b = dict(flat(torch.load(os.path.join(str(tmp_path), name), weights_only=False)))
        assert a.keys() == b.keys(), name
        for k, v in a.items():
            same = torch.equal(v, b[k]) if torch.is_tensor(v) else v == b[k]
            assert same, (name, k)
