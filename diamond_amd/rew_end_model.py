"""Reward / termination model (reference models/rew_end_model.py), inference path native.

`predict_rew_end` is on every WorldModelEnv.step (world_model_env.py:95-105) and in the
initial-condition burn-in (:123-124).  Encoder = the same fused conv / FiLM / attention
kernels as the denoiser at C = 32; the LSTM step is two MFMA GEMMs + one pointwise kernel.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import engine as E
from . import native as nv
from .blocks import Downsample, FilmTable, ResBlocks, RunCtx, conv3x3


@dataclass
class RewEndModelConfig:
    lstm_dim: int
    img_channels: int
    img_size: int
    cond_channels: int
    depths: List[int]
    channels: List[int]
    attn_depths: List[int]
    num_actions: Optional[int] = None


def init_lstm(model: nn.Module) -> None:
    """xavier / orthogonal / forget-bias-1 (reference utils.py:184-196)."""
    for name, p in model.named_parameters():
        if "weight_ih" in name:
            nn.init.xavier_uniform_(p.data)
        elif "weight_hh" in name:
            nn.init.orthogonal_(p.data)
        elif "bias_ih" in name:
            p.data.fill_(0)
            n = p.size(0)
            p.data[(n // 4):(n // 2)].fill_(1)
        elif "bias_hh" in name:
            p.data.fill_(0)


class RewEndEncoder(nn.Module):
    def __init__(self, in_channels: int, cond_channels: int, depths: List[int], channels: List[int],
                 attn_depths: List[int]) -> None:
        super().__init__()
        assert len(depths) == len(channels) == len(attn_depths)
        self.conv_in = conv3x3(in_channels, channels[0])
        blocks = []
        for i, n in enumerate(depths):
            c1, c2 = channels[max(0, i - 1)], channels[i]
            blocks.append(ResBlocks([c1] + [c2] * (n - 1), [c2] * n, cond_channels, bool(attn_depths[i])))
        blocks.append(ResBlocks([channels[-1]] * 2, [channels[-1]] * 2, cond_channels, True))
        self.blocks = nn.ModuleList(blocks)
        self.downsamples = nn.ModuleList([nn.Identity()] + [Downsample(c) for c in channels[:-1]] + [nn.Identity()])

    def run(self, ctx: RunCtx, x_nhwc16: Tensor) -> E.Act:
        x = E.conv2d([(E.Act(x_nhwc16), nv.PROLOGUE_NONE, None)], ctx.cache.conv_weight(self.conv_in),
                     ctx.cache.conv_bias(self.conv_in), self.conv_in.out_channels, naive=ctx.naive, w_f16=ctx.w16(self.conv_in))
        for blocks, down in zip(self.blocks, self.downsamples):
            if not isinstance(down, nn.Identity):
                x = down.run(ctx, x)
            x, _ = blocks.run(ctx, x)
        return x


class RewEndModel(nn.Module):
    def __init__(self, cfg: RewEndModelConfig) -> None:
        super().__init__()
        self.cfg = cfg
        self.encoder = RewEndEncoder(2 * cfg.img_channels, cfg.cond_channels, cfg.depths, cfg.channels, cfg.attn_depths)
        self.act_emb = nn.Embedding(cfg.num_actions, cfg.cond_channels)
        input_dim_lstm = cfg.channels[-1] * (cfg.img_size // 2 ** (len(cfg.depths) - 1)) ** 2
        self.lstm = nn.LSTM(input_dim_lstm, cfg.lstm_dim, batch_first=True)
        self.head = nn.Sequential(nn.Linear(cfg.lstm_dim, cfg.lstm_dim), nn.SiLU(), nn.Linear(cfg.lstm_dim, 3 + 2, bias=False))
        init_lstm(self.lstm)
        self._cache = E.PackCache()
        self._film: Optional[FilmTable] = None

    @property
    def device(self) -> torch.device:
        return self.act_emb.weight.device

    def _lstm_in_weight(self) -> Tensor:
        """weight_ih_l0 with its (e, h, w)-major columns (rew_end_model.py:52) permuted to the
        encoder's NHWC flatten order (h, w, e), so no activation transpose is needed."""
        def pack(w: Tensor) -> Tensor:
            e = self.cfg.channels[-1]
            s = int(round((w.shape[1] // e) ** 0.5))
            return w.detach().float().reshape(w.shape[0], e, s, s).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
        return self._cache.get(self.lstm.weight_ih_l0, "nhwc_cols", pack)

    @torch.no_grad()
    def predict_rew_end(self, obs: Tensor, act: Tensor, next_obs: Tensor,
                        hx_cx: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor, Tuple[Tensor, Tensor]]:
        b, t, c, h, w = obs.shape
        dev = obs.device
        x = torch.cat((obs.reshape(b * t, c, h, w), next_obs.reshape(b * t, c, h, w)), dim=1)
        x16 = E.nchw_to_nhwc(x, 16)
        cond = self._cache.f32(self.act_emb.weight)[act.reshape(b * t)].contiguous()  # embedding gather (plumbing)
        if self._film is None:
            self._film = FilmTable(self.encoder)
        ctx = RunCtx(self._cache, self._film, self._film.compute(cond))
        feat = self.encoder.run(ctx, x16).t.reshape(b, t, -1)  # NHWC flatten; weight columns permuted to match
        hd = self.cfg.lstm_dim
        if hx_cx is None:
            hx = torch.zeros(b, hd, device=dev)
            cx = torch.zeros(b, hd, device=dev)
        else:
            hx, cx = hx_cx[0][0].contiguous(), hx_cx[1][0].contiguous()
        w_ih, w_hh = self._lstm_in_weight(), self._cache.f32(self.lstm.weight_hh_l0)
        b_ih, b_hh = self._cache.f32(self.lstm.bias_ih_l0), self._cache.f32(self.lstm.bias_hh_l0)
        # input projections of all T steps in one GEMM, then the sequential recurrence
        gates_x = E.linear(feat.reshape(b * t, -1), w_ih, b_ih).reshape(b, t, 4 * hd)
        ys = []
        for i in range(t):
            g = gates_x[:, i].contiguous()
            E.linear(hx, w_hh, b_hh, out=g, accumulate=True)
            hn, cn = torch.empty_like(hx), torch.empty_like(cx)
            nv.check(nv.lib().dmd_lstm_pointwise(nv.fptr(g), nv.fptr(cx), nv.fptr(hn), nv.fptr(cn), b, hd, nv.stream()),
                     "dmd_lstm_pointwise")
            hx, cx = hn, cn
            ys.append(hx)
        y = torch.stack(ys, dim=1).reshape(b * t, hd)
        y = E.linear(y, self._cache.f32(self.head[0].weight), self._cache.f32(self.head[0].bias), silu=True)
        logits = E.linear(y, self._cache.f32(self.head[2].weight), None).reshape(b, t, -1)
        return logits[:, :, :-2], logits[:, :, -2:], (hx.unsqueeze(0), cx.unsqueeze(0))

    def forward(self, batch):
        raise NotImplementedError("RewEndModel.forward (training loss, reference rew_end_model.py:57-90) needs the encoder "
                                  "backward kernels: out of the imagined-rollout path (SURVEY.md §8f)")
