"""Generate golden fixtures by EXECUTING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py

Imports /root/reference/src (with the stubs of `_refimport.py`), overwrites the weights with
`diamond_amd.testing.fill_module_` (name-keyed, reproducible anywhere), runs the reference's
own Denoiser / DiffusionSampler / RewEndModel / ActorCritic / WorldModelEnv / env_loop on
seeded synthetic inputs and stores the outputs as small tensors in `tests/golden/*.pt`.
The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refimport as R  # noqa: E402

R.install()
from diamond_amd.testing import (fill_module_, initial_condition_batches, rew_end_train_batch, synthetic_actions,  # noqa: E402
                                 synthetic_frames)

torch.set_num_threads(8)
WEIGHT_SEED = 7


def ref_agent(num_actions=4, **kw):
    from agent import Agent

    agent = Agent(R.default_agent_config(num_actions=num_actions, **kw))
    fill_module_(agent, WEIGHT_SEED)
    return agent.eval()


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def gen_denoiser(agent, tag, h=64, w=64, b=2, only=None):
    """Denoiser.denoise at the three sampler sigmas + a per-sample (B,) sigma (`only`: indices to keep)."""
    from models.diffusion import DiffusionSampler, DiffusionSamplerConfig

    g = torch.Generator().manual_seed(11)
    obs = synthetic_frames(g, b, 12, h, w)
    act = synthetic_actions(g, 4, b, 4)
    noise = torch.randn(b, 3, h, w, generator=g)
    den = agent.denoiser
    sampler = DiffusionSampler(den, DiffusionSamplerConfig(num_steps_denoising=3))
    out = {"sigmas": sampler.sigmas.clone(), "seed": 11}
    with torch.no_grad():
        for i, sigma in enumerate(list(sampler.sigmas[:-1]) + [torch.tensor([0.7, 1.9][:b])]):
            if only is not None and i not in only:
                continue
            x = noise * sigma.reshape(-1, 1, 1, 1) + obs[:, -3:] * 0.5
            cs = den.compute_conditioners(sigma)
            f = den.compute_model_output(x, obs, act, cs)
            d = den.wrap_model_output(x, f, cs)
            d2 = den.denoise(x, sigma, obs, act)
            assert torch.equal(d, d2)
            out[f"model_output_{i}"] = f.clone()
            out[f"denoised_u8_{i}"] = d.add(1).div(2).mul(255).round().to(torch.uint8)
            if i == 0:
                im = den.inner_model
                out["cond_0"] = im.cond_proj(im.noise_emb(cs.c_noise) + im.act_emb(act)).clone()
    save(f"denoiser_{tag}.pt", out)


def gen_denoiser_pixels(agent, tag, h=64, w=64, b=17, sampled=3):
    """The quantised-frame budget (<= 1e-4 of the pixels on another uint8 level than the reference's) needs enough pixels to be
    resolved: b * 3 * h * w >= 200k, where 1e-4 is >= 20 pixels instead of 1-3.  Denoiser.denoise at the sampler's sigmas (the
    first `sampled` of them) and at a per-sample (B,) sigma; only the uint8 levels are stored (and one fp32 model output)."""
    from models.diffusion import DiffusionSampler, DiffusionSamplerConfig

    g = torch.Generator().manual_seed(31)
    obs = synthetic_frames(g, b, 12, h, w)
    act = synthetic_actions(g, 4, b, 4)
    noise = torch.randn(b, 3, h, w, generator=g)
    den = agent.denoiser
    sampler = DiffusionSampler(den, DiffusionSamplerConfig(num_steps_denoising=3))
    per_sample = torch.linspace(0.05, 4.0, b)
    out = {"sigmas": sampler.sigmas.clone(), "seed": 31, "b": b, "h": h, "w": w, "per_sample_sigma": per_sample, "pixels": b * 3 * h * w}
    with torch.no_grad():
        for i, sigma in enumerate(list(sampler.sigmas[:sampled]) + [per_sample]):
            x = noise * sigma.reshape(-1, 1, 1, 1) + obs[:, -3:] * 0.5
            d = den.denoise(x, sigma, obs, act)
            out[f"denoised_u8_{i}"] = d.add(1).div(2).mul(255).round().to(torch.uint8)
            if i == 0:
                out["model_output_0_sample0"] = den.compute_model_output(x, obs, act, den.compute_conditioners(sigma))[:1].clone()
    save(f"denoiser_pixels_{tag}.pt", out)


def gen_sampler(agent):
    from models.diffusion import DiffusionSampler, DiffusionSamplerConfig

    g = torch.Generator().manual_seed(12)
    out = {"seed": 12}
    for name, cfg, b in (
        ("euler3", DiffusionSamplerConfig(num_steps_denoising=3), 2),
        ("heun4", DiffusionSamplerConfig(num_steps_denoising=4, order=2), 1),
    ):
        prev_obs = synthetic_frames(g, b, 4, 3, 64, 64)
        prev_act = synthetic_actions(g, 4, b, 4)
        sampler = DiffusionSampler(agent.denoiser, cfg)
        seed = 100 + b
        torch.manual_seed(seed)  # consumed by torch.randn at diffusion_sampler.py:36
        x, traj = sampler.sample(prev_obs, prev_act)
        out[name] = {"noise_seed": seed, "x": x.clone(), "trajectory": torch.stack(traj, 1).clone(),
                     "sigmas": sampler.sigmas.clone()}
    save("sampler.pt", out)


def gen_sampler_heun5(agent):
    """BASELINE configs[3]'s sampler form (2nd-order Heun) at a batch of 2 over 5 steps (9 denoiser calls): the whole
    reference trajectory, so that every step can be teacher-forced."""
    from models.diffusion import DiffusionSampler, DiffusionSamplerConfig

    g = torch.Generator().manual_seed(21)
    b = 2
    prev_obs = synthetic_frames(g, b, 4, 3, 64, 64)
    prev_act = synthetic_actions(g, 4, b, 4)
    sampler = DiffusionSampler(agent.denoiser, DiffusionSamplerConfig(num_steps_denoising=5, order=2))
    torch.manual_seed(121)  # consumed by torch.randn at diffusion_sampler.py:36
    with torch.no_grad():
        x, traj = sampler.sample(prev_obs, prev_act)
    save("sampler_heun5.pt", {"seed": 21, "noise_seed": 121, "x": x.clone(), "trajectory": torch.stack(traj, 1).clone(),
                              "sigmas": sampler.sigmas.clone()})


def gen_sampler_heun_pixels(agent, b=17, steps=(1, 3)):
    """configs[3]'s 2nd-order Heun step where the pixel budget can be resolved: b * 3 * 64 * 64 >= 200k values per step.  Of the
    reference's 5-step Heun trajectory at batch b only the points around `steps` are kept (x_i -> x_{i+1}: two denoiser
    evaluations each, teacher-forced by the test), fp32."""
    from models.diffusion import DiffusionSampler, DiffusionSamplerConfig

    g = torch.Generator().manual_seed(23)
    prev_obs = synthetic_frames(g, b, 4, 3, 64, 64)
    prev_act = synthetic_actions(g, 4, b, 4)
    sampler = DiffusionSampler(agent.denoiser, DiffusionSamplerConfig(num_steps_denoising=5, order=2))
    torch.manual_seed(123)
    with torch.no_grad():
        _, traj = sampler.sample(prev_obs, prev_act)
    out = {"seed": 23, "noise_seed": 123, "b": b, "sigmas": sampler.sigmas.clone(), "steps": list(steps), "pixels": b * 3 * 64 * 64}
    for i in steps:
        out[f"x_{i}"], out[f"x_{i + 1}"] = traj[i].clone(), traj[i + 1].clone()
    save("sampler_heun_pixels.pt", out)


def gen_rew_end(agent, name="rew_end.pt", size=64):
    g = torch.Generator().manual_seed(13)
    b = 2
    obs = synthetic_frames(g, b, 4, 3, size, size)
    act = synthetic_actions(g, 4, b, 4)
    m = agent.rew_end_model
    with torch.no_grad():
        lr, le, (hx, cx) = m.predict_rew_end(obs[:, :-1], act[:, :-1], obs[:, 1:])  # burn-in form, T=3
        lr2, le2, (hx2, cx2) = m.predict_rew_end(obs[:, -1:], act[:, -1:], obs[:, :1], (hx, cx))  # step form
    save(name, {"seed": 13, "logits_rew": lr, "logits_end": le, "hx": hx, "cx": cx,
                        "logits_rew_step": lr2, "logits_end_step": le2, "hx_step": hx2, "cx_step": cx2})


def gen_actor_critic(agent, name="actor_critic.pt", size=64):
    g = torch.Generator().manual_seed(14)
    b = 3
    ac = agent.actor_critic
    obs = synthetic_frames(g, b, 3, size, size)
    obs2 = synthetic_frames(g, b, 3, size, size)
    hx = torch.randn(b, 512, generator=g) * 0.3
    cx = torch.randn(b, 512, generator=g) * 0.3
    ac.zero_grad()
    o1 = ac.predict_act_value(obs, (hx, cx))
    o2 = ac.predict_act_value(obs2, o1.hx_cx)
    w = torch.randn(b, 4, generator=g)
    loss = (o2.logits_act * w).sum() + o2.val.square().sum() + o1.val.sum() + 0.1 * o2.hx_cx[1].sum()
    loss.backward()
    grads = {k: p.grad.clone() for k, p in ac.named_parameters()}
    save(name, {
        "seed": 14, "logits1": o1.logits_act.detach(), "val1": o1.val.detach(), "logits2": o2.logits_act.detach(),
        "val2": o2.val.detach(), "hx2": o2.hx_cx[0].detach(), "cx2": o2.hx_cx[1].detach(), "loss": loss.detach(),
        # norms accumulated in fp64: an fp32 norm over the 2M-element LSTM matrices is itself only good to ~1e-4
        "grad_norms": {k: v.double().norm() for k, v in grads.items()},
        "grads_small": {k: v for k, v in grads.items() if v.numel() <= 20000},
    })


class _FakeLoader:
    """What WorldModelEnv needs from a DataLoader (world_model_env.py:38,115-122)."""

    class _BS:
        def __init__(self, b):
            self.batch_size = b

    def __init__(self, batch, seed, size=64):
        self.batch_sampler = self._BS(batch)
        self._batch, self._seed, self._size = batch, seed, size

    def __iter__(self):
        from data import Batch

        for obs, act in initial_condition_batches(self._seed, self._batch, 4, h=self._size, w=self._size):
            yield Batch(obs=obs, act=act, rew=None, end=None, trunc=None, mask_padding=None, info=None, segment_ids=None)


def gen_window(name="window.pt", size=64, b=4, horizon=4, t=6, agent=None):
    """Two BPTT windows of ActorCritic.forward()+backward through the reference's own
    WorldModelEnv and env_loop, default RNG seeded (draw order: SURVEY App. A.5)."""
    from envs import WorldModelEnv, WorldModelEnvConfig
    from models.actor_critic import ActorCriticLossConfig
    from models.diffusion import DiffusionSamplerConfig, SigmaDistributionConfig

    agent = agent or ref_agent(img_size=size)
    env = WorldModelEnv(agent.denoiser, agent.rew_end_model, _FakeLoader(b, seed=21, size=size),
                        WorldModelEnvConfig(horizon=horizon, num_batches_to_preload=2,
                                            diffusion_sampler=DiffusionSamplerConfig(num_steps_denoising=3)))
    agent.setup_training(SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         ActorCriticLossConfig(backup_every=t, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                               weight_entropy_loss=0.001), env)
    torch.manual_seed(1234)
    random.seed(0)
    out = {"b": b, "horizon": horizon, "backup_every": t, "pool_seed": 21, "rng_seed": 1234, "preload": 2, "size": size, "windows": []}
    ac = agent.actor_critic
    for _ in range(2):
        ac.zero_grad()
        # peek at the rollout the loss is built from, by re-running forward() pieces: we call
        # env_loop directly (same as actor_critic.py:77) and then the loss code path via forward
        # is not re-runnable on the same generator state, so replicate lines 79-88 here.
        all_obs, act, rew, end, trunc, logits_act, val, val_bootstrap, _ = ac.env_loop.send(t)
        from torch.distributions.categorical import Categorical
        import torch.nn.functional as F
        from models.actor_critic import compute_lambda_returns

        c = ac.loss_cfg
        d = Categorical(logits=logits_act)
        entropy = d.entropy().mean()
        lam = compute_lambda_returns(rew, end, trunc, val_bootstrap, c.gamma, c.lambda_)
        loss = (-d.log_prob(act) * (lam - val).detach()).mean() + c.weight_value_loss * F.mse_loss(val, lam) \
            - c.weight_entropy_loss * entropy
        loss.backward()
        out["windows"].append({
            "obs_u8": all_obs.add(1).div(2).mul(255).round().to(torch.uint8), "act": act, "rew": rew, "end": end,
            "trunc": trunc, "logits_act": logits_act.detach(), "val": val.detach(), "val_bootstrap": val_bootstrap,
            "lambda_returns": lam, "loss": loss.detach(), "entropy": entropy.detach(),
            "grad_norms": {k: p.grad.norm().clone() for k, p in ac.named_parameters()},
        })
        print("window: loss", float(loss), "ends", int(end.sum()), "truncs", int(trunc.sum()))
    save(name, out)


def gen_window_teacher_forced():
    """The same two windows as gen_window(), with everything the reference's env handed to env_loop recorded per
    step (observations as uint8 levels -- they are on the 256-level grid up to one ulp --, rewards, ends, truncs,
    final observations, burn-in frames).  tests/test_gpu_models.py replays that stream into this repo's
    env_loop + ActorCritic (teacher forcing: no quantisation flip can feed back), which pins logits / values /
    loss / gradients of a whole window at the 1e-4 tolerance."""
    from envs import WorldModelEnv, WorldModelEnvConfig
    from models.actor_critic import ActorCriticLossConfig, compute_lambda_returns
    from models.diffusion import DiffusionSamplerConfig, SigmaDistributionConfig
    from torch.distributions.categorical import Categorical
    import torch.nn.functional as F

    u8 = lambda x: x.add(1).div(2).mul(255).round().to(torch.uint8)
    agent = ref_agent()
    b, horizon, t = 4, 4, 6
    env = WorldModelEnv(agent.denoiser, agent.rew_end_model, _FakeLoader(b, seed=21),
                        WorldModelEnvConfig(horizon=horizon, num_batches_to_preload=2,
                                            diffusion_sampler=DiffusionSamplerConfig(num_steps_denoising=3)))
    log = {"reset_obs_u8": None, "steps": []}
    orig_reset, orig_step = env.reset, env.step

    def reset(**kw):
        obs, info = orig_reset(**kw)
        log["reset_obs_u8"] = u8(obs)
        return obs, info

    def step(act):
        obs, rew, end, trunc, info = orig_step(act)
        rec = {"obs_u8": u8(obs), "rew": rew.clone(), "end": end.clone(), "trunc": trunc.clone(), "act": act.clone()}
        if "final_observation" in info:
            rec["final_observation_u8"] = u8(info["final_observation"])
            rec["burnin_obs_u8"] = u8(info["burnin_obs"])
        log["steps"].append(rec)
        return obs, rew, end, trunc, info

    env.reset, env.step = reset, step
    agent.setup_training(SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         ActorCriticLossConfig(backup_every=t, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                               weight_entropy_loss=0.001), env)
    torch.manual_seed(1234)
    random.seed(0)
    out = {"b": b, "horizon": horizon, "backup_every": t, "rng_seed": 1234, "windows": []}
    ac = agent.actor_critic
    for _ in range(2):
        ac.zero_grad()
        n0 = len(log["steps"])
        all_obs, act, rew, end, trunc, logits_act, val, val_bootstrap, _ = ac.env_loop.send(t)
        c = ac.loss_cfg
        d = Categorical(logits=logits_act)
        lam = compute_lambda_returns(rew, end, trunc, val_bootstrap, c.gamma, c.lambda_)
        loss = (-d.log_prob(act) * (lam - val).detach()).mean() + c.weight_value_loss * F.mse_loss(val, lam) \
            - c.weight_entropy_loss * d.entropy().mean()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in ac.named_parameters()}
        out["windows"].append({
            "steps": log["steps"][n0:], "act": act, "logits_act": logits_act.detach(), "val": val.detach(),
            "val_bootstrap": val_bootstrap, "loss": loss.detach(),
            "grad_norms": {k: v.double().norm() for k, v in grads.items()},  # fp64 accumulation
            # whole tensors up to 40k elements (all conv / GroupNorm / head parameters), every 97th element of the LSTM matrices
            "grads": {k: (v if v.numel() <= 40000 else v.flatten()[::97].clone()) for k, v in grads.items()},
        })
        print("teacher-forced window: loss", float(loss), "resets", sum("burnin_obs_u8" in s for s in log["steps"][n0:]))
    out["reset_obs_u8"] = log["reset_obs_u8"]
    save("window_tf.pt", out)


def gen_denoiser_train():
    """Denoiser.forward (the denoising TRAINING loss with autoregressive context refresh, denoiser.py:93-122) +
    loss.backward() of the reference: loss and every parameter gradient."""
    from data import Batch
    from models.diffusion import SigmaDistributionConfig

    agent = ref_agent()
    den = agent.denoiser
    den.setup_training(SigmaDistributionConfig(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20))
    g = torch.Generator().manual_seed(31)
    b, t = 2, 6  # 4 conditioning frames + 2 predicted ones
    obs = synthetic_frames(g, b, t, 3, 64, 64)
    act = synthetic_actions(g, 4, b, t)
    mask = torch.ones(b, t, dtype=torch.bool)
    mask[1, 5] = False  # a padded step: excluded from the loss of the second prediction
    batch = Batch(obs=obs, act=act, rew=None, end=None, trunc=None, mask_padding=mask, info=None, segment_ids=None)
    den.zero_grad()
    torch.manual_seed(77)  # consumed by sample_sigma / apply_noise (denoiser.py:55,62-63)
    loss, logs = den(batch)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in den.named_parameters()}
    assert all(v is not None for v in grads.values())
    save("denoiser_train.pt", {
        "seed": 31, "rng_seed": 77, "b": b, "t": t, "loss": loss.detach(),
        "grad_norms": {k: v.double().norm() for k, v in grads.items()},
        "grads": {k: (v if v.numel() <= 4096 else v.flatten()[::13].clone()) for k, v in grads.items()},  # every 13th element of the larger tensors
    })
    print("denoiser training step: loss", float(loss))


def gen_rew_end_train():
    """RewEndModel.forward (reward / termination cross-entropies over a segment, rew_end_model.py:57-90) +
    loss.backward() of the reference: losses, confusion matrices and every parameter gradient."""
    from data import Batch

    agent = ref_agent()
    m = agent.rew_end_model
    g = torch.Generator().manual_seed(41)
    batch = Batch(**rew_end_train_batch(g))
    m.zero_grad()
    loss, logs = m(batch)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(v is not None for v in grads.values())
    save("rew_end_train.pt", {
        "seed": 41, "loss": loss.detach(), "loss_rew": logs["loss_rew"], "loss_end": logs["loss_end"],
        "cm_rew": logs["confusion_matrix"]["rew"], "cm_end": logs["confusion_matrix"]["end"],
        "grad_norms": {k: v.double().norm() for k, v in grads.items()},
        "grads": {k: (v if v.numel() <= 4096 else v.flatten()[::13].clone()) for k, v in grads.items()},
    })
    print("rew/end training step: loss", float(loss), "cm_rew", logs["confusion_matrix"]["rew"].tolist())


def gen_offgrid_72():
    """RewEndModel / ActorCritic at 72x72 (levels 72/36/18/9 and 72/36/18/9/4: off the kernels' 8-pixel tile grid; the
    reference runs any size its strides divide, rew_end_model.py:33 / actor_critic.py:45)"""
    agent = ref_agent(img_size=72)
    gen_rew_end(agent, "rew_end_72x72.pt", 72)
    gen_actor_critic(agent, "actor_critic_72x72.pt", 72)
    gen_window("window_72x72.pt", size=72, b=3, horizon=3, t=4)  # whole path: sampler, reward / end model, resets, AC fwd + bwd


def gen_wide():
    """Configurations wider than the default one (tests/wide_configs.py): the reference's Denoiser (model output, training loss and
    gradients), RewEndModel and ActorCritic (forward + gradients) as standalone modules, name-keyed weights."""
    from data import Batch
    from models.actor_critic import ActorCritic, ActorCriticConfig
    from models.diffusion import Denoiser, DenoiserConfig, InnerModelConfig, SigmaDistributionConfig
    from models.rew_end_model import RewEndModel, RewEndModelConfig
    from tests import wide_configs as W

    s = W.SIZE
    out = {}
    # -- denoiser: inference
    den = Denoiser(DenoiserConfig(inner_model=InnerModelConfig(**W.DENOISER), sigma_data=0.5, sigma_offset_noise=0.3))
    fill_module_(den, W.WEIGHT_SEED)
    den.eval()
    g = torch.Generator().manual_seed(5)
    obs, act, x = synthetic_frames(g, 2, 12, s, s), synthetic_actions(g, 4, 2, 4), torch.randn(2, 3, s, s, generator=g)
    with torch.no_grad():
        for i, sigma in enumerate((torch.tensor(0.7), torch.tensor([0.05, 3.0]))):
            out[f"model_output_{i}"] = den.compute_model_output(x, obs, act, den.compute_conditioners(sigma)).clone()
    # -- denoiser: training step
    den.train()
    den.setup_training(SigmaDistributionConfig(**W.SIGMA_DIST))
    g = torch.Generator().manual_seed(31)
    obs, act = synthetic_frames(g, 1, 5, 3, s, s), synthetic_actions(g, 4, 1, 5)
    den.zero_grad()
    torch.manual_seed(77)
    loss, _ = den(Batch(obs=obs, act=act, rew=None, end=None, trunc=None, mask_padding=torch.ones(1, 5, dtype=torch.bool), info=None,
                        segment_ids=None))
    loss.backward()
    out["train"] = {"loss": loss.detach().clone(), "grad_norms": {k: p.grad.double().norm() for k, p in den.named_parameters()},
                    "grads": {k: W.sample_grad(p.grad) for k, p in den.named_parameters()}}
    # -- reward / end model
    m = RewEndModel(RewEndModelConfig(**W.REW_END))
    fill_module_(m, W.WEIGHT_SEED + 1)
    m.eval()
    g = torch.Generator().manual_seed(9)
    obs, act = synthetic_frames(g, 2, 3, 3, s, s), synthetic_actions(g, 4, 2, 2)
    with torch.no_grad():
        lr, le, (h, c) = m.predict_rew_end(obs[:, :-1], act, obs[:, 1:])
    out["rew_end"] = {"logits_rew": lr.clone(), "logits_end": le.clone(), "h": h.clone(), "c": c.clone()}
    # -- actor-critic: forward + backward
    ac = ActorCritic(ActorCriticConfig(**W.ACTOR_CRITIC))
    fill_module_(ac, W.WEIGHT_SEED + 2)
    g = torch.Generator().manual_seed(11)
    obs = synthetic_frames(g, 2, 3, s, s)
    o = ac.predict_act_value(obs, None)
    (o.logits_act.square().sum() + o.val.sum()).backward()
    out["actor_critic"] = {"logits": o.logits_act.detach().clone(), "val": o.val.detach().clone(),
                           "grad_norms": {k: p.grad.double().norm() for k, p in ac.named_parameters()},
                           "grads": {k: W.sample_grad(p.grad) for k, p in ac.named_parameters()}}
    save("wide.pt", out)


def ref_wide_agent():
    """the reference's Agent on the wide configurations of tests/wide_configs.py"""
    from agent import Agent, AgentConfig
    from models.actor_critic import ActorCriticConfig
    from models.diffusion import DenoiserConfig, InnerModelConfig
    from models.rew_end_model import RewEndModelConfig
    from tests import wide_configs as W

    agent = Agent(W.agent_config(AgentConfig, DenoiserConfig, InnerModelConfig, RewEndModelConfig, ActorCriticConfig))
    fill_module_(agent, WEIGHT_SEED)
    return agent.eval()


def main():
    if "--wide" in sys.argv:
        gen_wide()
        return
    if "--wide-window" in sys.argv:  # the whole path (sampler, reward / end model, resets, actor-critic forward + backward) on the wide networks
        from tests import wide_configs as W

        gen_window("window_wide.pt", size=W.SIZE, b=3, horizon=3, t=4, agent=ref_wide_agent())
        return
    if "--offgrid" in sys.argv:
        gen_offgrid_72()
        return
    if "--pixels" in sys.argv:  # >= 200k pixels per quantised-frame check
        gen_denoiser_pixels(ref_agent(), "default")
        gen_denoiser_pixels(ref_agent(), "72x72", h=72, w=72, b=14, sampled=1)
        gen_denoiser_pixels(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011", b=17, sampled=1)
        gen_sampler_heun_pixels(ref_agent())
        return
    if "--heun-pixels" in sys.argv:
        gen_sampler_heun_pixels(ref_agent())
        return
    if "--rew-end-train" in sys.argv:
        gen_rew_end_train()
        return
    if "--denoiser-train" in sys.argv:
        gen_denoiser_train()
        return
    if "--sampler-heun5" in sys.argv:
        gen_sampler_heun5(ref_agent())
        return
    if "--denoiser-ragged" in sys.argv:
        # sizes whose U-Net levels are not multiples of the kernels' tiles: 72x72 (levels 72 / 36 / 18 / 9: the reference pads
        # nothing) and 68x76 (padded to 72x80 inside UNet.forward and cropped back, blocks.py:227-229,247); the second one also
        # with attention at the two deepest levels
        gen_denoiser(ref_agent(), "72x72", h=72, w=72, b=2, only=(0, 3))
        gen_denoiser(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011_68x76", h=68, w=76, b=1, only=(1, 3))
        return
    if "--denoiser-256" in sys.argv:  # BASELINE configs[4] shape: one 256x256 frame, attention at the two deepest levels
        gen_denoiser(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011_256", h=256, w=256, b=1, only=(1, 3))
        return
    if "--window-tf" in sys.argv or "--actor-critic" in sys.argv:  # (re)generate single fixtures
        if "--window-tf" in sys.argv:
            gen_window_teacher_forced()
        if "--actor-critic" in sys.argv:
            gen_actor_critic(ref_agent())
        return
    agent = ref_agent()
    save("state_dict_keys.pt", {k: tuple(v.shape) for k, v in agent.state_dict().items()})
    gen_denoiser(agent, "default")
    gen_sampler(agent)
    gen_sampler_heun5(agent)
    gen_rew_end(agent)
    gen_actor_critic(agent)
    gen_window()
    gen_window_teacher_forced()
    gen_denoiser_train()
    gen_rew_end_train()
    # attention at 16x16 and 8x8 inside the U-Net (BASELINE config 5 uses attn_depths=[0,0,1,1])
    gen_denoiser(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011", b=1)
    gen_denoiser(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011_256", h=256, w=256, b=1, only=(1, 3))
    gen_denoiser(ref_agent(), "72x72", h=72, w=72, b=2, only=(0, 3))
    gen_denoiser(ref_agent(denoiser_attn_depths=(0, 0, 1, 1)), "attn0011_68x76", h=68, w=76, b=1, only=(1, 3))


if __name__ == "__main__":
    main()
