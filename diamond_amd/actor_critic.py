"""Actor-critic (reference models/actor_critic.py): conv encoder + LSTMCell + heads, the
REINFORCE-with-baseline loss over a 15-step imagined rollout, and lambda-returns.

Execution: the conv encoder (all of the model's convolution/GroupNorm/SiLU/pooling work)
runs on hand-written HIP kernels, forward AND backward, behind one torch.autograd.Function
(diamond_amd/ac_native.py); the LSTM cell and the two linear heads likewise (MFMA GEMMs `dmd_linear`,
`dmd_lstm_pointwise(_bwd)`, diamond_amd/lstm_native.py), one Function per policy step chained by torch for BPTT.
There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import math
from collections import namedtuple
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import native as nv
from .blocks import SmallResBlock, conv3x3
from .env_loop import make_env_loop
from .rew_end_model import init_lstm

ActorCriticOutput = namedtuple("ActorCriticOutput", "logits_act val hx_cx")


@dataclass
class ActorCriticLossConfig:
    backup_every: int
    gamma: float
    lambda_: float
    weight_value_loss: float
    weight_entropy_loss: float


@dataclass
class ActorCriticConfig:
    lstm_dim: int
    img_channels: int
    img_size: int
    channels: List[int]
    down: List[int]
    num_actions: Optional[int] = None


class ActorCriticEncoder(nn.Module):
    def __init__(self, cfg: ActorCriticConfig) -> None:
        super().__init__()
        assert len(cfg.channels) == len(cfg.down)
        layers: List[nn.Module] = [conv3x3(cfg.img_channels, cfg.channels[0])]
        for i, ch in enumerate(cfg.channels):
            layers.append(SmallResBlock(cfg.channels[max(0, i - 1)], ch))
            if cfg.down[i]:
                layers.append(nn.MaxPool2d(2))
        self.encoder = nn.Sequential(*layers)


class ActorCritic(nn.Module):
    def __init__(self, cfg: ActorCriticConfig) -> None:
        super().__init__()
        self.cfg = cfg
        self.encoder = ActorCriticEncoder(cfg)
        self.lstm_dim = cfg.lstm_dim
        input_dim_lstm = cfg.channels[-1] * (cfg.img_size // 2 ** (sum(cfg.down))) ** 2
        self.lstm = nn.LSTMCell(input_dim_lstm, cfg.lstm_dim)
        self.critic_linear = nn.Linear(cfg.lstm_dim, 1)
        self.actor_linear = nn.Linear(cfg.lstm_dim, cfg.num_actions)
        for lin in (self.actor_linear, self.critic_linear):
            lin.weight.data.fill_(0)
            lin.bias.data.fill_(0)
        init_lstm(self.lstm)
        self.env_loop = None
        self.loss_cfg = None
        self.backend = "native"  # conv encoder fwd+bwd on libdiamond_hip (diamond_amd/ac_native.py)
        self._native_encoder = None
        self.expo_fn = None  # test hook: injected exponential draws for action sampling

    @property
    def device(self) -> torch.device:
        return self.lstm.weight_hh.device

    def setup_training(self, rl_env, loss_cfg: ActorCriticLossConfig) -> None:
        assert self.env_loop is None and self.loss_cfg is None
        self.env_loop = make_env_loop(rl_env, self, expo_fn=lambda l: None if self.expo_fn is None else self.expo_fn(l))
        self.loss_cfg = loss_cfg

    # -- evaluation ----------------------------------------------------------------------
    def encode(self, obs: Tensor) -> Tensor:
        """Flattened (c, h, w) encoder features (reference :70-71) on the HIP kernels."""
        if self._native_encoder is None:
            from .ac_native import NativeEncoder
            self._native_encoder = NativeEncoder(self.encoder.encoder)
        nv.check_current_device(obs.device)  # (ctypes launches go to the CURRENT device's stream: a policy on another GPU raises)
        return self._native_encoder(obs)

    def predict_act_value(self, obs: Tensor, hx_cx: Optional[Tuple[Tensor, Tensor]]) -> ActorCriticOutput:
        assert obs.ndim == 4
        if hx_cx is None:
            z = obs.new_zeros(obs.size(0), self.lstm_dim)
            hx_cx = (z, z)
        hx, cx = hx_cx
        x = self.encode(obs)
        # nn.LSTMCell (gate order i, f, g, o) + the two heads (reference :72-73): dmd_linear / dmd_lstm_pointwise
        # forward and backward (lstm_native.LstmHeadsFn)
        from .lstm_native import lstm_heads
        logits, val, hx, cx = lstm_heads(self._native_encoder.cache, x, hx, cx, self.lstm, self.actor_linear, self.critic_linear)
        return ActorCriticOutput(logits, val, (hx, cx))

    def predict_from_features(self, x: Tensor, hx_cx: Tuple[Tensor, Tensor]) -> ActorCriticOutput:
        """predict_act_value behind the encoder: LSTM cell + heads on features `encode` produced (env_loop encodes the burn-in
        frames of a reset in ONE pass and steps the LSTM over them)."""
        from .lstm_native import lstm_heads
        nv.check_current_device(x.device)
        hx, cx = hx_cx
        logits, val, hx, cx = lstm_heads(self._native_encoder.cache, x, hx, cx, self.lstm, self.actor_linear, self.critic_linear)
        return ActorCriticOutput(logits, val, (hx, cx))

    def burn_in_from_features(self, x: Tensor, num_frames: int) -> Tuple[Tensor, Tensor]:
        """(hx, cx) of the LSTM stepped from the zero state over `num_frames` frames' features x (num_frames * k, F),
        frame-major: the policy-side burn-in of a reset (reference env_loop.py:53-56) as one autograd node
        (lstm_native.LstmBurnInFn); bitwise `num_frames` calls of predict_from_features."""
        from .lstm_native import lstm_burn_in
        return lstm_burn_in(self._native_encoder.cache, x, num_frames, self.lstm)

    def forward(self):
        c = self.loss_cfg
        _, act, rew, end, trunc, logits_act, val, val_bootstrap, _ = self.env_loop.send(c.backup_every)
        return actor_critic_loss(logits_act, val, act, rew, end, trunc, val_bootstrap, c)


def actor_critic_loss(logits_act, val, act, rew, end, trunc, val_bootstrap, c: ActorCriticLossConfig):
    """Policy gradient with lambda-return baseline + value MSE - entropy bonus
    (reference actor_critic.py:79-96); tiny (B,T,A) tensors, plain torch."""
    logp_all = logits_act - logits_act.logsumexp(dim=-1, keepdim=True)
    entropy = -(logp_all.exp() * logp_all).sum(-1).mean()
    logp = logp_all.gather(-1, act.unsqueeze(-1)).squeeze(-1)
    lambda_returns = compute_lambda_returns(rew, end, trunc, val_bootstrap, c.gamma, c.lambda_)
    loss_actions = (-logp * (lambda_returns - val).detach()).mean()
    loss_values = c.weight_value_loss * F.mse_loss(val, lambda_returns)
    loss_entropy = -c.weight_entropy_loss * entropy
    loss = loss_actions + loss_entropy + loss_values
    metrics = {
        "policy_entropy": entropy.detach() / math.log(2),
        "loss_actions": loss_actions.detach(),
        "loss_entropy": loss_entropy.detach(),
        "loss_values": loss_values.detach(),
        "loss_total": loss.detach(),
    }
    return loss, metrics


@torch.no_grad()
def compute_lambda_returns(rew: Tensor, end: Tensor, trunc: Tensor, val_bootstrap: Tensor, gamma: float,
                           lambda_: float) -> Tensor:
    """TD(lambda) targets with sign-clipped rewards (reference actor_critic.py:116-143)."""
    assert rew.ndim == 2 and rew.size() == end.size() == trunc.size() == val_bootstrap.size()
    rew = rew.sign()
    alive = (end + trunc).clip(max=1).logical_not()
    ret = rew + (1 - end) * gamma * ((1 - trunc) * (1 - lambda_) + trunc) * val_bootstrap
    if lambda_ == 0:
        return ret
    last = val_bootstrap[:, -1]
    for t in reversed(range(rew.size(1))):
        ret[:, t] += alive[:, t] * gamma * lambda_ * last
        last = ret[:, t]
    return ret
