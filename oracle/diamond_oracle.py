"""CPU ORACLE for the DIAMOND imagined-rollout hot path.  *** TEST INFRASTRUCTURE ONLY ***

This file is a from-scratch, functional (state-dict driven, no nn.Module) restatement of
the reference algorithm, used exclusively as the checker in `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.  The product
(`diamond_amd/`) never imports it; the product path has no CPU fallback.

Pinning status: the reference ships no tests / golden vectors.  This oracle
is pinned against outputs of the reference itself, executed in the build container by
`tests/golden/make_golden.py` (fixtures committed under `tests/golden/`); see
`tests/test_oracle_golden.py`.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src).  Arithmetic is plain torch on CPU in the dtype of the inputs, so the
same code run in float64 gives the "truth" used to show that the HIP path is not further
from exact arithmetic than the fp32 CPU path is.

Conventions: `sd` is a flat dict of tensors keyed exactly like the reference
`state_dict()` of the sub-model (e.g. for the denoiser: "inner_model.conv_in.weight").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

GN_GROUP_SIZE = 32  # models/blocks.py:12
GN_EPS = 1e-5  # models/blocks.py:13
ATTN_HEAD_DIM = 8  # models/blocks.py:14

SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# configs (values = config/agent/default.yaml:1-32, config/trainer.yaml:66-80,134-140)
# --------------------------------------------------------------------------------------
@dataclass
class DenoiserSpec:
    img_channels: int = 3
    num_steps_conditioning: int = 4
    cond_channels: int = 256
    depths: Sequence[int] = (2, 2, 2, 2)
    channels: Sequence[int] = (64, 64, 64, 64)
    attn_depths: Sequence[int] = (0, 0, 0, 0)
    sigma_data: float = 0.5
    sigma_offset_noise: float = 0.3


@dataclass
class SamplerSpec:
    num_steps_denoising: int = 3
    sigma_min: float = 2e-3
    sigma_max: float = 5.0
    rho: int = 7
    order: int = 1
    s_churn: float = 0.0
    s_tmin: float = 0.0
    s_tmax: float = float("inf")
    s_noise: float = 1.0


@dataclass
class ActorCriticSpec:
    lstm_dim: int = 512
    img_channels: int = 3
    img_size: int = 64
    channels: Sequence[int] = (32, 32, 64, 64)
    down: Sequence[int] = (1, 1, 1, 1)


@dataclass
class RewEndSpec:
    lstm_dim: int = 512
    img_channels: int = 3
    img_size: int = 64
    cond_channels: int = 128
    depths: Sequence[int] = (2, 2, 2, 2)
    channels: Sequence[int] = (32, 32, 32, 32)
    attn_depths: Sequence[int] = (0, 0, 0, 0)


@dataclass
class LossSpec:
    backup_every: int = 15
    gamma: float = 0.985
    lambda_: float = 0.95
    weight_value_loss: float = 1.0
    weight_entropy_loss: float = 0.001


# --------------------------------------------------------------------------------------
# building blocks (models/blocks.py)
# --------------------------------------------------------------------------------------
def silu(x: Tensor) -> Tensor:
    """F.silu = x * sigmoid(x) (blocks.py:143-144, 119)."""
    return x * torch.sigmoid(x)


def group_norm(x: Tensor, num_groups: int, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None) -> Tensor:
    """Per-(sample, group) normalisation with biased variance, eps inside the rsqrt
    (blocks.py:28-31 affine form, blocks.py:43 non-affine form)."""
    n, c, h, w = x.shape
    xg = x.reshape(n, num_groups, -1)
    mean = xg.mean(dim=-1, keepdim=True)
    var = (xg - mean).square().mean(dim=-1, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + GN_EPS)).reshape(n, c, h, w)
    if weight is not None:
        y = y * weight.view(1, c, 1, 1) + bias.view(1, c, 1, 1)
    return y


def num_groups(c: int) -> int:
    return max(1, c // GN_GROUP_SIZE)  # blocks.py:27,38


def ada_group_norm(sd: SD, p: str, x: Tensor, cond: Tensor) -> Tensor:
    """AdaGroupNorm.forward blocks.py:41-45: GN (no affine) then x*(1+scale)+shift with
    [scale | shift] = Linear(cond); the first half of the channels is the scale."""
    c = x.shape[1]
    y = group_norm(x, num_groups(c))
    ss = F.linear(cond, sd[p + ".linear.weight"], sd[p + ".linear.bias"])
    scale, shift = ss[:, :c, None, None], ss[:, c:, None, None]
    return y * (1 + scale) + shift


def conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def self_attention_2d(sd: SD, p: str, x: Tensor) -> Tensor:
    """SelfAttention2d.forward blocks.py:62-72.  Heads are 8 contiguous channels; q,k,v are
    the three channel thirds of the 1x1 projection; scale 1/sqrt(8) after QK^T; the
    residual is taken on the *normalised* input."""
    n, c, h, w = x.shape
    heads = max(1, c // ATTN_HEAD_DIM)
    d = c // heads
    xn = group_norm(x, num_groups(c), sd[p + ".norm.norm.weight"], sd[p + ".norm.norm.bias"])
    qkv = conv(sd, p + ".qkv_proj", xn, padding=0)  # (n, 3c, h, w)
    qkv = qkv.reshape(n, 3, heads, d, h * w).permute(0, 1, 2, 4, 3)  # (n, 3, heads, hw, d)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    att = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(d), dim=-1)
    y = (att @ v).transpose(2, 3).reshape(n, c, h, w)
    return xn + conv(sd, p + ".out_proj", y, padding=0)


def res_block(sd: SD, p: str, x: Tensor, cond: Tensor, attn: bool) -> Tensor:
    """ResBlock.forward blocks.py:141-147."""
    r = conv(sd, p + ".proj", x, padding=0) if (p + ".proj.weight") in sd else x
    h = conv(sd, p + ".conv1", silu(ada_group_norm(sd, p + ".norm1", x, cond)))
    h = conv(sd, p + ".conv2", silu(ada_group_norm(sd, p + ".norm2", h, cond)))
    h = h + r
    if attn:
        h = self_attention_2d(sd, p + ".attn", h)
    return h


def res_blocks(sd: SD, p: str, n: int, x: Tensor, cond: Tensor, attn: bool, to_cat: Optional[List[Tensor]] = None):
    """ResBlocks.forward blocks.py:171-177 (cat puts the running x first)."""
    outs = []
    for i in range(n):
        if to_cat is not None:
            x = torch.cat((x, to_cat[i]), dim=1)
        x = res_block(sd, f"{p}.resblocks.{i}", x, cond, attn)
        outs.append(x)
    return x, outs


def unet(sd: SD, p: str, x: Tensor, cond: Tensor, depths: Sequence[int], attn_depths: Sequence[int]) -> Tensor:
    """UNet.forward blocks.py:224-246.  u_blocks / upsamples are stored deepest-first
    (blocks.py:210,219-220); level i>0 is entered through a stride-2 conv, left through
    nearest-x2 + conv; each up level has depth+1 blocks consuming the reversed skips."""
    L = len(depths)
    h, w = x.shape[-2:]
    m = 2 ** (L - 1)
    x = F.pad(x, (0, math.ceil(w / m) * m - w, 0, math.ceil(h / m) * m - h))  # blocks.py:227-229
    skips = []
    for i in range(L):
        if i > 0:
            x = conv(sd, f"{p}.downsamples.{i}.conv", x, stride=2)
        x_down = x
        x, outs = res_blocks(sd, f"{p}.d_blocks.{i}", depths[i], x, cond, bool(attn_depths[i]))
        skips.append([x_down] + outs)
    x, _ = res_blocks(sd, f"{p}.mid_blocks", 2, x, cond, True)
    for j in range(L):
        lvl = L - 1 - j
        if j > 0:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv(sd, f"{p}.upsamples.{j}.conv", x)
        x, _ = res_blocks(sd, f"{p}.u_blocks.{j}", depths[lvl] + 1, x, cond, bool(attn_depths[lvl]), skips[lvl][::-1])
    return x[..., :h, :w]


# --------------------------------------------------------------------------------------
# denoiser (models/diffusion/inner_model.py, denoiser.py)
# --------------------------------------------------------------------------------------
def fourier_features(weight: Tensor, c_noise: Tensor) -> Tensor:
    """FourierFeatures.forward blocks.py:84-87."""
    f = 2 * math.pi * c_noise.unsqueeze(1) @ weight
    return torch.cat([f.cos(), f.sin()], dim=-1)


def cond_vector(sd: SD, c_noise: Tensor, act: Tensor) -> Tensor:
    """inner_model.py:45 — cond_proj(noise_emb(c_noise) + flatten(act_emb(act)))."""
    e = fourier_features(sd["inner_model.noise_emb.weight"], c_noise)
    a = F.embedding(act, sd["inner_model.act_emb.0.weight"]).flatten(1)
    y = F.linear(e + a, sd["inner_model.cond_proj.0.weight"], sd["inner_model.cond_proj.0.bias"])
    return F.linear(silu(y), sd["inner_model.cond_proj.2.weight"], sd["inner_model.cond_proj.2.bias"])


def inner_model(sd: SD, spec: DenoiserSpec, noisy: Tensor, c_noise: Tensor, obs: Tensor, act: Tensor) -> Tensor:
    """InnerModel.forward inner_model.py:44-49."""
    cond = cond_vector(sd, c_noise, act)
    x = conv(sd, "inner_model.conv_in", torch.cat((obs, noisy), dim=1))
    x = unet(sd, "inner_model.unet", x, cond, spec.depths, spec.attn_depths)
    c = x.shape[1]
    x = group_norm(x, num_groups(c), sd["inner_model.norm_out.norm.weight"], sd["inner_model.norm_out.norm.bias"])
    return conv(sd, "inner_model.conv_out", silu(x))


def conditioners(spec: DenoiserSpec, sigma: Tensor):
    """compute_conditioners denoiser.py:66-72 (c_in,c_out,c_skip get 4 dims, c_noise 1)."""
    s = (sigma ** 2 + spec.sigma_offset_noise ** 2).sqrt()
    c_in = 1 / (s ** 2 + spec.sigma_data ** 2).sqrt()
    c_skip = spec.sigma_data ** 2 / (s ** 2 + spec.sigma_data ** 2)
    c_out = s * c_skip.sqrt()
    c_noise = s.log() / 4
    d4 = lambda t: t.reshape(t.shape + (1,) * (4 - t.ndim))
    d1 = lambda t: t.reshape(t.shape + (1,) * (1 - t.ndim))
    return d4(c_in), d4(c_out), d4(c_skip), d1(c_noise)


def model_output(sd: SD, spec: DenoiserSpec, noisy: Tensor, sigma: Tensor, obs: Tensor, act: Tensor) -> Tensor:
    """compute_model_output denoiser.py:74-77 (pre-quantisation network output F)."""
    c_in, _, _, c_noise = conditioners(spec, sigma)
    return inner_model(sd, spec, noisy * c_in, c_noise, obs / spec.sigma_data, act)


def quantize_frame(d: Tensor) -> Tensor:
    """wrap_model_output denoiser.py:83 — clamp, map to {0..255} by truncation, map back."""
    return d.clamp(-1, 1).add(1).div(2).mul(255).byte().to(d.dtype).div(255).mul(2).sub(1)


def denoise(sd: SD, spec: DenoiserSpec, noisy: Tensor, sigma: Tensor, obs: Tensor, act: Tensor,
            return_model_output: bool = False):
    """Denoiser.denoise denoiser.py:86-91."""
    _, c_out, c_skip, _ = conditioners(spec, sigma)
    f = model_output(sd, spec, noisy, sigma, obs, act)
    d = quantize_frame(c_skip * noisy + c_out * f)
    return (d, f) if return_model_output else d


# --------------------------------------------------------------------------------------
# sampler (models/diffusion/diffusion_sampler.py)
# --------------------------------------------------------------------------------------
def build_sigmas(spec: SamplerSpec, dtype=torch.float32) -> Tensor:
    """build_sigmas diffusion_sampler.py:61-66 (Karras schedule + trailing 0)."""
    lo = spec.sigma_min ** (1 / spec.rho)
    hi = spec.sigma_max ** (1 / spec.rho)
    l = torch.linspace(0, 1, spec.num_steps_denoising, dtype=dtype)
    s = (hi + l * (lo - hi)) ** spec.rho
    return torch.cat((s, s.new_zeros(1)))


def sample(sd: SD, dspec: DenoiserSpec, sspec: SamplerSpec, prev_obs: Tensor, prev_act: Tensor, noise: Tensor,
           churn_noise: Optional[Callable[[Tensor], Tensor]] = None,
           denoise_fn: Optional[Callable] = None, sigmas: Optional[Tensor] = None) -> Tuple[Tensor, List[Tensor]]:
    """DiffusionSampler.sample diffusion_sampler.py:30-58.  `noise` is the injected
    x0 ~ N(0,1) draw (line 36); `churn_noise(x)` supplies randn_like draws (line 42).
    `denoise_fn(x, sigma, obs, act)` may replace the oracle denoiser (teacher forcing); `sigmas` may replace the
    schedule (a two-element slice of it = ONE step from a teacher-forced trajectory point passed as `noise`)."""
    b, t, c, h, w = prev_obs.shape
    obs = prev_obs.reshape(b, t * c, h, w)
    sigmas = build_sigmas(sspec, prev_obs.dtype) if sigmas is None else sigmas
    dn = denoise_fn or (lambda x, s, o, a: denoise(sd, dspec, x, s, o, a))
    s_in = torch.ones(b, dtype=prev_obs.dtype)
    gamma_ = min(sspec.s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1)
    x = noise
    traj = [x]
    for sigma, nxt in zip(sigmas[:-1], sigmas[1:]):
        gamma = gamma_ if sspec.s_tmin <= sigma <= sspec.s_tmax else 0
        sigma_hat = sigma * (gamma + 1)
        if gamma > 0:
            x = x + churn_noise(x) * sspec.s_noise * (sigma_hat ** 2 - sigma ** 2) ** 0.5
        den = dn(x, sigma, obs, prev_act)
        d = (x - den) / sigma_hat
        dt = nxt - sigma_hat
        if sspec.order == 1 or nxt == 0:
            x = x + d * dt  # Euler, line 49
        else:  # Heun, lines 52-56 (second evaluation gets a (B,) sigma)
            x2 = x + d * dt
            den2 = dn(x2, nxt * s_in, obs, prev_act)
            d2 = (x2 - den2) / nxt
            x = x + (d + d2) / 2 * dt
        traj.append(x)
    return x, traj


# --------------------------------------------------------------------------------------
# categorical sampling (torch.distributions.Categorical as used at env_loop.py:32,
# world_model_env.py:103-104)
# --------------------------------------------------------------------------------------
def categorical_sample(logits: Tensor, e: Tensor) -> Tensor:
    """Categorical(logits).sample() == argmax(softmax(logits) / E) with
    E = empty_like(probs).exponential_(1) drawn from the default generator.
    `e` is that injected draw."""
    p = torch.softmax(logits - logits.logsumexp(dim=-1, keepdim=True), dim=-1)
    return (p / e).argmax(dim=-1)


def categorical_entropy_logprob(logits: Tensor, act: Tensor):
    lp = logits - logits.logsumexp(dim=-1, keepdim=True)
    p = lp.exp()
    ent = -(p * lp).sum(-1)
    return ent, lp.gather(-1, act.unsqueeze(-1)).squeeze(-1)


# --------------------------------------------------------------------------------------
# LSTM cell (nn.LSTMCell / one step of nn.LSTM; gate order i, f, g, o)
# --------------------------------------------------------------------------------------
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor):
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


# --------------------------------------------------------------------------------------
# actor-critic (models/actor_critic.py)
# --------------------------------------------------------------------------------------
def ac_encoder(sd: SD, spec: ActorCriticSpec, x: Tensor, pool_choice=None, tie_gaps=None) -> Tensor:
    """ActorCriticEncoder actor_critic.py:101-113 + SmallResBlock blocks.py:116-123.

    pool_choice (test aid): one (N, C, Ho, Wo) tensor per MaxPool2d with the element (2 dy + dx) of each 2x2 window to
    take INSTEAD of this function's own argmax -- teacher-forces the discrete pooling decisions of another run, so that
    gradients can be compared where two window elements agree to within rounding (a tie broken the other way re-routes a
    gradient entry: a finite difference that no tolerance on smooth arithmetic covers).  tie_gaps (list) receives, per
    pooling layer, max over windows of (own max - chosen element) / max|x|: the caller asserts these are rounding-size."""
    x = conv(sd, "encoder.encoder.0", x)
    idx = 1
    npool = 0
    for i, ch in enumerate(spec.channels):
        cin = spec.channels[max(0, i - 1)]
        p = f"encoder.encoder.{idx}"
        y = group_norm(x, num_groups(cin), sd[p + ".f.0.norm.weight"], sd[p + ".f.0.norm.bias"])
        y = conv(sd, p + ".f.2", silu(y))
        skip = conv(sd, p + ".skip_projection", x, padding=0) if cin != ch else x
        x = skip + y
        idx += 1
        if spec.down[i]:
            if pool_choice is None:
                x = F.max_pool2d(x, 2)
            else:
                n, c, h, w = x.shape
                win = x.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
                chosen = win.gather(4, pool_choice[npool].long().unsqueeze(4)).squeeze(4)
                if tie_gaps is not None:
                    tie_gaps.append(float(((win.amax(4) - chosen).abs().max() / x.abs().max()).detach()))
                x = chosen
                npool += 1
            idx += 1
    return x


def ac_predict(sd: SD, spec: ActorCriticSpec, obs: Tensor, hx: Tensor, cx: Tensor):
    """ActorCritic.predict_act_value actor_critic.py:68-73 (flatten is (c,h,w)-major)."""
    x = ac_encoder(sd, spec, obs).flatten(1)
    hx, cx = lstm_cell(x, hx, cx, sd["lstm.weight_ih"], sd["lstm.weight_hh"], sd["lstm.bias_ih"], sd["lstm.bias_hh"])
    logits = F.linear(hx, sd["actor_linear.weight"], sd["actor_linear.bias"])
    val = F.linear(hx, sd["critic_linear.weight"], sd["critic_linear.bias"]).squeeze(1)
    return logits, val, (hx, cx)


def lambda_returns(rew, end, trunc, val_bootstrap, gamma, lambda_):
    """compute_lambda_returns actor_critic.py:116-143."""
    rew = rew.sign()
    end_or_trunc = (end + trunc).clip(max=1)
    ret = rew + (1 - end) * gamma * ((1 - trunc) * (1 - lambda_) + trunc) * val_bootstrap
    if lambda_ == 0:
        return ret
    last = val_bootstrap[:, -1]
    for t in reversed(range(rew.size(1))):
        ret[:, t] += end_or_trunc[:, t].logical_not() * gamma * lambda_ * last
        last = ret[:, t]
    return ret


def ac_loss(logits_act, val, act, rew, end, trunc, val_bootstrap, lc: LossSpec):
    """ActorCritic.forward actor_critic.py:79-96 (everything after the rollout)."""
    ent, logp = categorical_entropy_logprob(logits_act, act)
    entropy = ent.mean()
    with torch.no_grad():
        lam = lambda_returns(rew, end, trunc, val_bootstrap, lc.gamma, lc.lambda_)
    loss_actions = (-logp * (lam - val).detach()).mean()
    loss_values = lc.weight_value_loss * F.mse_loss(val, lam)
    loss_entropy = -lc.weight_entropy_loss * entropy
    loss = loss_actions + loss_entropy + loss_values
    metrics = {
        "policy_entropy": entropy.detach() / math.log(2),
        "loss_actions": loss_actions.detach(),
        "loss_entropy": loss_entropy.detach(),
        "loss_values": loss_values.detach(),
        "loss_total": loss.detach(),
    }
    return loss, metrics


# --------------------------------------------------------------------------------------
# reward / end model (models/rew_end_model.py)
# --------------------------------------------------------------------------------------
def rew_end_encoder(sd: SD, spec: RewEndSpec, x: Tensor, cond: Tensor) -> Tensor:
    """RewEndEncoder.forward rew_end_model.py:128-133 (blocks 0..L-1 + attention block L)."""
    L = len(spec.depths)
    x = conv(sd, "encoder.conv_in", x)
    for i in range(L):
        if i > 0:
            x = conv(sd, f"encoder.downsamples.{i}.conv", x, stride=2)
        x, _ = res_blocks(sd, f"encoder.blocks.{i}", spec.depths[i], x, cond, bool(spec.attn_depths[i]))
    x, _ = res_blocks(sd, f"encoder.blocks.{L}", 2, x, cond, True)
    return x


def rew_end_predict(sd: SD, spec: RewEndSpec, obs: Tensor, act: Tensor, next_obs: Tensor, hx_cx=None):
    """RewEndModel.predict_rew_end rew_end_model.py:42-55.  obs/next_obs (B,T,C,H,W),
    act (B,T); returns logits_rew (B,T,3), logits_end (B,T,2), (h, c) each (1,B,512)."""
    b, t, c, h, w = obs.shape
    x = torch.cat((obs.reshape(b * t, c, h, w), next_obs.reshape(b * t, c, h, w)), dim=1)
    cond = F.embedding(act.reshape(b * t), sd["act_emb.weight"])
    x = rew_end_encoder(sd, spec, x, cond).reshape(b, t, -1)
    if hx_cx is None:
        hx = x.new_zeros(b, spec.lstm_dim)
        cx = x.new_zeros(b, spec.lstm_dim)
    else:
        hx, cx = hx_cx[0][0], hx_cx[1][0]
    ys = []
    for i in range(t):
        hx, cx = lstm_cell(x[:, i], hx, cx, sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"],
                           sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])
        ys.append(hx)
    y = torch.stack(ys, dim=1)
    y = F.linear(silu(F.linear(y, sd["head.0.weight"], sd["head.0.bias"])), sd["head.2.weight"])
    return y[:, :, :-2], y[:, :, -2:], (hx.unsqueeze(0), cx.unsqueeze(0))


# --------------------------------------------------------------------------------------
# imagination environment + rollout driver with injected randomness
# (envs/world_model_env.py:25-139, coroutines/env_loop.py:12-74)
# --------------------------------------------------------------------------------------
class DrawSource:
    """Injected randomness in the reference's consumption order:
    per imagined step  E_act (B,A) -> randn (B,3,H,W) -> E_rew (B,3) -> E_end (B,2)."""

    def __init__(self, generator: torch.Generator, dtype=torch.float32):
        self.g = generator
        self.dtype = dtype

    def exponential(self, *shape) -> Tensor:
        return torch.empty(*shape, dtype=torch.float32).exponential_(1, generator=self.g).to(self.dtype)

    def randn(self, *shape) -> Tensor:
        return torch.randn(*shape, generator=self.g, dtype=torch.float32).to(self.dtype)


@dataclass
class AgentSD:
    denoiser: SD
    rew_end_model: SD
    actor_critic: SD
    dspec: DenoiserSpec = field(default_factory=DenoiserSpec)
    sspec: SamplerSpec = field(default_factory=SamplerSpec)
    aspec: ActorCriticSpec = field(default_factory=ActorCriticSpec)
    rspec: RewEndSpec = field(default_factory=RewEndSpec)


class ImaginationEnv:
    """WorldModelEnv restated (world_model_env.py:45-105).  `pool` is an iterator of
    (obs (B,T,C,H,W), act (B,T)) initial-condition batches; the rew/end LSTM is burnt in on
    the first T-1 transitions exactly as make_generator_init does (lines 119-129), and
    dead envs are refilled sample-by-sample in pool order (lines 131-139)."""

    def __init__(self, agent: AgentSD, pool_batches, num_envs: int, horizon: int, draws: DrawSource,
                 num_batches_to_preload: int = 1):
        self.a = agent
        self.num_envs = num_envs
        self.horizon = horizon
        self.draws = draws
        self._it = iter(pool_batches)
        self._preload = num_batches_to_preload
        self._pool: List[Tuple[Tensor, Tensor, Tensor, Tensor]] = []
        self._cursor = 0

    def _refill(self):
        self._pool, self._cursor = [], 0
        for _ in range(self._preload):
            obs, act = next(self._it)
            with torch.no_grad():
                *_, (hx, cx) = rew_end_predict(self.a.rew_end_model, self.a.rspec, obs[:, :-1], act[:, :-1], obs[:, 1:])
            for i in range(obs.shape[0]):
                self._pool.append((obs[i], act[i], hx[0, i], cx[0, i]))

    def _take(self, n: int):
        # world_model_env.py:133 — a request that does not fit in the remainder of the pool
        # discards the remainder and preloads a fresh pool.
        if self._cursor + n > len(self._pool):
            self._refill()
        items = self._pool[self._cursor:self._cursor + n]
        self._cursor += n
        obs = torch.stack([i[0] for i in items])
        act = torch.stack([i[1] for i in items])
        hx = torch.stack([i[2] for i in items]).unsqueeze(0)
        cx = torch.stack([i[3] for i in items]).unsqueeze(0)
        return obs, act, hx, cx

    @torch.no_grad()
    def reset(self):
        self.obs_buffer, self.act_buffer, self.hx, self.cx = self._take(self.num_envs)
        self.ep_len = torch.zeros(self.num_envs, dtype=torch.long)
        return self.obs_buffer[:, -1]

    @torch.no_grad()
    def step(self, act: Tensor):
        a = self.a
        self.act_buffer[:, -1] = act
        b, t, c, h, w = self.obs_buffer.shape
        next_obs, traj = sample(a.denoiser, a.dspec, a.sspec, self.obs_buffer, self.act_buffer,
                                self.draws.randn(b, c, h, w))
        lr, le, (self.hx, self.cx) = rew_end_predict(a.rew_end_model, a.rspec, self.obs_buffer[:, -1:],
                                                    self.act_buffer[:, -1:], next_obs.unsqueeze(1), (self.hx, self.cx))
        rew = categorical_sample(lr, self.draws.exponential(b, 1, 3)).squeeze(1) - 1.0
        end = categorical_sample(le, self.draws.exponential(b, 1, 2)).squeeze(1)
        self.ep_len += 1
        trunc = (self.ep_len >= self.horizon).long()
        self.obs_buffer = self.obs_buffer.roll(-1, dims=1)
        self.act_buffer = self.act_buffer.roll(-1, dims=1)
        self.obs_buffer[:, -1] = next_obs
        dead = torch.logical_or(end, trunc)
        info = {"denoising_trajectory": torch.stack(traj, dim=1)}
        if dead.any():
            o, ac, hx, cx = self._take(int(dead.sum()))
            self.obs_buffer[dead] = o
            self.act_buffer[dead] = ac
            self.hx[:, dead] = hx
            self.cx[:, dead] = cx
            self.ep_len[dead] = 0
            info["final_observation"] = next_obs[dead]
            info["burnin_obs"] = self.obs_buffer[dead, :-1]
        return self.obs_buffer[:, -1], rew, end, trunc, info


def rollout(agent: AgentSD, env: ImaginationEnv, state, num_steps: int, draws: DrawSource):
    """One BPTT window of make_env_loop (env_loop.py:26-74).  `state` = (obs, hx, cx) carried
    between windows (hx/cx are detached at the window start, line 27).  Returns the stacked
    (B,T,...) tensors the actor-critic loss consumes, and the new state."""
    obs, hx, cx = state
    hx, cx = hx.detach(), cx.detach()
    sd, sp = agent.actor_critic, agent.aspec
    rows, dead, val_final = [], None, None
    for n in range(num_steps):
        logits, val, (hx, cx) = ac_predict(sd, sp, obs, hx, cx)
        num_actions = logits.shape[-1]
        act = categorical_sample(logits.detach(), draws.exponential(obs.shape[0], num_actions))
        next_obs, rew, end, trunc, info = env.step(act)
        if n > 0:
            vb = val.detach().clone()
            if dead.any():
                vb[dead] = val_final
            rows[-1][-1] = vb
        dead = torch.logical_or(end, trunc)
        if dead.any():
            with torch.no_grad():
                _, val_final, _ = ac_predict(sd, sp, info["final_observation"], hx[dead], cx[dead])
            gate = 1 - dead.float().unsqueeze(1)
            hx, cx = hx * gate, cx * gate
            if "burnin_obs" in info:
                bo = info["burnin_obs"]
                for i in range(bo.size(1)):
                    _, _, (h_new, c_new) = ac_predict(sd, sp, bo[:, i], hx[dead], cx[dead])
                    hx = hx.clone()
                    cx = cx.clone()
                    hx[dead], cx[dead] = h_new, c_new
        rows.append([obs, act, rew, end, trunc, logits, val, None])
        obs = next_obs
    with torch.no_grad():
        _, vb, _ = ac_predict(sd, sp, obs, hx, cx)
    if dead.any():
        vb[dead] = val_final
    rows[-1][-1] = vb
    cols = [torch.stack(c, dim=1) for c in zip(*rows)]
    return cols, (obs, hx, cx)
